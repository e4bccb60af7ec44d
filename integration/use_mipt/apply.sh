#!/bin/sh
# integration/use_mipt/apply.sh <dir> — adds the USE_MIPT switch to a checkout (or scratch copy) of nbonneel/pathtracer in <dir>.
#
# What a maintainer commits is seven small edits, every one behind `#ifdef USE_MIPT` (the precedent is the reference's own
# USE_EMBREE switch, Geometry.cpp:602-682); the code they pull in is the four .inc files beside this script.  The edits are
# made by anchor line, each anchor must exist exactly once, and the script changes nothing when any of them is missing.
# No text of the reference is stored in this repository: the anchors below are the only strings of it that appear here.
#
#   Raytracer.h      + #include "mipt.h"; members of class Raytracer          (raytracer_members.inc)
#   Raytracer.cpp    the two sample loops keep their bodies as render_image_cpu / render_image_nopreviz_cpu;
#                    + the binding at the end of the file                         (raytracer_binding.inc)
#   Geometry.h       + members of class Scene                                      (scene_members.inc)
#   Geometry.cpp     Scene::intersection starts with the resident-scene shortcut
#   TriangleMesh.cpp TriMesh::build_bvh starts with the device build              (trimesh_build_bvh.inc)
#
# Build afterwards with  -DUSE_MIPT -I<repo>/include -I<repo>/integration/use_mipt  and link  -lmipt.
set -eu
D=${1:?usage: apply.sh <reference directory>}

anchor() {   # anchor <file> <fixed string>: must match exactly one whole line
	n=$(grep -c -x -F -- "$2" "$D/$1" || true)
	if [ "$n" != 1 ]; then echo "apply.sh: anchor not found exactly once in $1 (found $n): $2" >&2; exit 1; fi
}
A_RT_H_INC='#include "PointSet.h"'
A_RT_H_MEM='	void render_image_nopreviz();'
A_RT_C_IMG='void Raytracer::render_image()'
A_RT_C_NOP='void Raytracer::render_image_nopreviz() {'
A_GEO_H='	std::vector<TriMesh*> castToMesh; //horrible hack to avoid dynamic_casts on the fly'
A_GEO_C='bool Scene::intersection(const Ray& d, Vector& P, int &sphere_id, float &min_t, MaterialValues &mat, int &triangle_id, bool avoid_ghosts, bool isCoherent) const {'
A_TRI_C='void TriMesh::build_bvh(BVH* b, int i0, int i1) {'
anchor Raytracer.h "$A_RT_H_INC"; anchor Raytracer.h "$A_RT_H_MEM"
anchor Raytracer.cpp "$A_RT_C_IMG"; anchor Raytracer.cpp "$A_RT_C_NOP"
anchor Geometry.h "$A_GEO_H"; anchor Geometry.cpp "$A_GEO_C"; anchor TriangleMesh.cpp "$A_TRI_C"
if grep -q USE_MIPT "$D/Raytracer.h"; then echo "apply.sh: $D already carries USE_MIPT" >&2; exit 1; fi

# edit <file> <anchor> before|after|replace <text>: awk, whole-line fixed-string match (no regular expressions on the reference's text)
edit() {
	A="$2" HOW="$3" T="$4" awk 'BEGIN { a = ENVIRON["A"]; how = ENVIRON["HOW"]; t = ENVIRON["T"] }
		{ if ($0 == a) { if (how == "before") { print t; print } else if (how == "after") { print; print t } else print t } else print }' "$D/$1" > "$D/$1.mipt" && mv "$D/$1.mipt" "$D/$1"
}
edit Raytracer.h "$A_RT_H_INC" after '#ifdef USE_MIPT
#include "mipt.h"
#endif'
edit Raytracer.h "$A_RT_H_MEM" after '#ifdef USE_MIPT
#include "raytracer_members.inc"
#endif'
edit Raytracer.cpp "$A_RT_C_IMG" replace '#ifdef USE_MIPT
void Raytracer::render_image_cpu()
#else
void Raytracer::render_image()
#endif'
edit Raytracer.cpp "$A_RT_C_NOP" replace '#ifdef USE_MIPT
void Raytracer::render_image_nopreviz_cpu() {
#else
void Raytracer::render_image_nopreviz() {
#endif'
printf '%s\n' '#ifdef USE_MIPT' '#include "raytracer_binding.inc"' '#endif' >> "$D/Raytracer.cpp"
edit Geometry.h "$A_GEO_H" after '#ifdef USE_MIPT
#include "scene_members.inc"
#endif'
edit Geometry.cpp "$A_GEO_C" after '#ifdef USE_MIPT
	if (mipt_resident && !avoid_ghosts && !omp_in_parallel()) return intersection_mipt(d, P, sphere_id, min_t, mat, triangle_id);
#endif'
edit TriangleMesh.cpp "$A_TRI_C" before '#ifdef USE_MIPT
#include "trimesh_build_bvh.inc"
#endif'
edit TriangleMesh.cpp "$A_TRI_C" after '#ifdef USE_MIPT
	if (mipt_build_bvh_on_device(this, i0, i1)) return;
#endif'
echo "apply.sh: USE_MIPT switch added to $D"
