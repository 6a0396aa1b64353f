#include "obj_reader.hpp"

#include <cstdlib>
#include <cstring>
#include <fstream>
#include <stdexcept>

namespace rthost {

namespace {

struct P3 { float v[3]; };
struct P2 { float v[2]; };

// resolves a 1-based (or negative, relative) OBJ index; 0 = absent
inline long resolve(long idx, size_t count) { return idx > 0 ? idx - 1 : (idx < 0 ? (long)count + idx : -1); }

} // namespace

std::vector<MeshData> ReadObjFile(const std::string &path) {
    std::ifstream in(path);
    if (!in)
        throw std::runtime_error("Load Failed: cannot open " + path);
    std::vector<P3> pos, nrm;
    std::vector<P2> uv;
    std::vector<MeshData> meshes(1);
    std::string line;
    while (std::getline(in, line)) {
        const char *s = line.c_str();
        while (*s == ' ' || *s == '\t') s++;
        if (s[0] == 'v' && (s[1] == ' ' || s[1] == '\t')) {
            P3 p; char *e = const_cast<char *>(s + 1);
            for (int k = 0; k < 3; k++) p.v[k] = std::strtof(e, &e);
            pos.push_back(p);
        } else if (s[0] == 'v' && s[1] == 'n') {
            P3 p; char *e = const_cast<char *>(s + 2);
            for (int k = 0; k < 3; k++) p.v[k] = std::strtof(e, &e);
            nrm.push_back(p);
        } else if (s[0] == 'v' && s[1] == 't') {
            P2 p; char *e = const_cast<char *>(s + 2);
            for (int k = 0; k < 2; k++) p.v[k] = std::strtof(e, &e);
            uv.push_back(p);
        } else if ((s[0] == 'o' || s[0] == 'g') && (s[1] == ' ' || s[1] == '\t' || s[1] == 0)) {
            if (!meshes.back().indices.empty())
                meshes.emplace_back();
        } else if (s[0] == 'f' && (s[1] == ' ' || s[1] == '\t')) {
            MeshData &m = meshes.back();
            std::vector<unsigned int> corner;
            char *e = const_cast<char *>(s + 1);
            for (;;) {
                while (*e == ' ' || *e == '\t' || *e == '\r') e++;
                if (*e == 0)
                    break;
                long vi = std::strtol(e, &e, 10), ti = 0, ni = 0;
                if (*e == '/') {
                    e++;
                    if (*e != '/')
                        ti = std::strtol(e, &e, 10);
                    if (*e == '/') {
                        e++;
                        ni = std::strtol(e, &e, 10);
                    }
                }
                SdVertex v;
                std::memset(&v, 0, sizeof(v));
                long p = resolve(vi, pos.size()), t = resolve(ti, uv.size()), n = resolve(ni, nrm.size());
                if (p < 0 || (size_t)p >= pos.size())
                    throw std::runtime_error("Load Failed: bad vertex index in " + path);
                std::memcpy(v.position, pos[p].v, sizeof(v.position));
                if (n >= 0 && (size_t)n < nrm.size())
                    std::memcpy(v.normal, nrm[n].v, sizeof(v.normal));
                if (t >= 0 && (size_t)t < uv.size())
                    std::memcpy(v.texCoord, uv[t].v, sizeof(v.texCoord)); // else (0,0), ModelLoader.hpp:220-221
                corner.push_back((unsigned int)m.vertices.size());
                m.vertices.push_back(v);
            }
            for (size_t i = 1; i + 1 < corner.size(); i++) {
                m.indices.push_back(corner[0]);
                m.indices.push_back(corner[i]);
                m.indices.push_back(corner[i + 1]);
            }
        }
    }
    if (meshes.back().indices.empty())
        meshes.pop_back();
    if (meshes.empty())
        throw std::runtime_error("Load Failed: no faces in " + path);
    return meshes;
}

} // namespace rthost
