// obj_reader.hpp -- minimal Wavefront OBJ reader standing in for the reference's assimp import
// (Include/ModelLoader.hpp:263-268: Triangulate | JoinIdenticalVertices | CalcTangentSpace |
// SortByPType).  assimp is a third-party dependency absent from the reference tree and un-pinned, so
// only its observable output for the bundled files is reproduced: one mesh per object/group with
// faces, faces in file order, per-corner position/normal/uv, uv = 0 when the file has none, polygons
// with more than three corners split as a fan around their first corner.  Vertices are not merged
// (JoinIdenticalVertices changes only the index arrays, not the triangles the tracer sees).
#pragma once

#include <string>
#include <vector>

#include "sd_scene.hpp"

namespace rthost {

// Throws std::runtime_error("Load Failed: ...") when the file cannot be read or holds no faces.
std::vector<MeshData> ReadObjFile(const std::string &path);

} // namespace rthost
