#!/usr/bin/env python3
"""Per-launch digest of tools/pmc_levels.py output (csv on stdin); counters that were not collected are left out."""
import csv
import sys
for r in csv.DictReader(sys.stdin):
    def f(k):
        return float(r[k]) if r.get(k) not in (None, "") else None
    g = f('GRBM_GUI_ACTIVE') / 8
    iv, tc, slots = f('SQ_INSTS_VALU'), f('SQ_THREAD_CYCLES_VALU'), f('SQ_ACTIVE_INST_VALU')
    if g < 20000:
        continue
    out = [r['launch'], 'us=%.0f' % (g / 2.4e3), 'valu=%.3g' % iv]
    # lanes at work per issue slot: both counters weigh an instruction by the slots it takes (tools/lane_calibration.sh)
    if tc is not None and slots: out.append('lanes=%.1f' % (tc / slots))
    for k, name in (('SQ_INSTS_VMEM_RD', 'vmem'), ('SQ_INSTS_SALU', 'salu'), ('SQ_INSTS_LDS', 'lds'), ('SQ_LDS_BANK_CONFLICT', 'lds_conflict_cyc')):
        if f(k) is not None: out.append('%s=%.3g' % (name, f(k)))
    if f('TA_TA_BUSY_sum') is not None: out.append('TA=%.2f' % (f('TA_TA_BUSY_sum') / 256 / g))
    # issue slots per SIMD and clock; this GPU was measured to issue at most 0.39 (tools/valu_probe.hip)
    if slots: out.append('valu_slots_per_simd_clk=%.3f' % (slots / 1024 / g))
    if f('SQ_WAVES') is not None: out.append('waves=%.0f' % f('SQ_WAVES'))
    if f('SQ_WAVE_CYCLES') is not None: out.append('occ=%.1f' % (f('SQ_WAVE_CYCLES') / 4 / 1024 / g))
    if f('SQ_WAIT_INST_ANY') is not None and f('SQ_WAVE_CYCLES'): out.append('wait=%.2f' % (f('SQ_WAIT_INST_ANY') / f('SQ_WAVE_CYCLES')))
    if f('SQ_WAIT_INST_LDS') is not None and f('SQ_WAVES'): out.append('wait_lds_per_wave=%.3g' % (f('SQ_WAIT_INST_LDS') / f('SQ_WAVES')))
    if f('TCP_PENDING_STALL_CYCLES_sum') is not None: out.append('tcp_stall=%.2f' % (f('TCP_PENDING_STALL_CYCLES_sum') / 256 / g))
    if f('TCC_HIT_sum') is not None: out.append('l2_hit=%.3f' % (f('TCC_HIT_sum') / max(f('TCC_REQ_sum'), 1.0)))
    if f('TCP_TCC_READ_REQ_sum') is not None: out.append('l1_hit=%.3f' % (1.0 - f('TCP_TCC_READ_REQ_sum') / max(f('TCP_TOTAL_CACHE_ACCESSES_sum'), 1.0)))
    if f('FETCH_SIZE') is not None: out.append('hbm_read_MB=%.1f' % (f('FETCH_SIZE') / 1024.0))
    if f('WRITE_SIZE') is not None: out.append('hbm_write_MB=%.1f' % (f('WRITE_SIZE') / 1024.0))
    print(' '.join(out))
