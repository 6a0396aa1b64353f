#!/usr/bin/env python3
"""Per-launch digest of tools/pmc_levels.py output (csv on stdin)."""
import csv
import sys
for r in csv.DictReader(sys.stdin):
    g = float(r['GRBM_GUI_ACTIVE']) / 8
    iv = float(r['SQ_INSTS_VALU']); tc = float(r['SQ_THREAD_CYCLES_VALU'])
    if g < 20000:
        continue
    print(r['launch'], 'us=%.0f' % (g / 2.4e3), 'valu=%.3g' % iv, 'lanes=%.1f' % (tc / iv if iv else 0), 'vmem=%.3g' % float(r['SQ_INSTS_VMEM_RD']),
          'salu=%.3g' % float(r['SQ_INSTS_SALU']), 'TA=%.2f' % (float(r['TA_TA_BUSY_sum']) / 256 / g), 'valu_util=%.2f' % (iv * 4 / 1024 / g),
          'waves=%s' % r['SQ_WAVES'], 'occ=%.1f' % (float(r['SQ_WAVE_CYCLES']) / 4 / 1024 / g), 'tcp_stall=%.2f' % (float(r['TCP_PENDING_STALL_CYCLES_sum']) / 256 / g),
          'l2_hit=%.3f' % (float(r['TCC_HIT_sum']) / max(float(r['TCC_REQ_sum']), 1.0)), 'l1_hit=%.3f' % (1.0 - float(r['TCP_TCC_READ_REQ_sum']) / max(float(r['TCP_TOTAL_CACHE_ACCESSES_sum']), 1.0)),
          'hbm_read_MB=%.1f' % (float(r['FETCH_SIZE']) / 1024.0), 'hbm_write_MB=%.1f' % (float(r['WRITE_SIZE']) / 1024.0))
