# usage: tools/build_variant.sh <name> <extra hipcc flags...>  -> pathtracer_amd/libmipt_<name>.so (tuning builds; select with MIPT_LIB_OVERRIDE)
name=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -pthread -ffp-contract=off -fPIC -shared -std=c++17 -Wno-unused-value "$@" -o pathtracer_amd/libmipt_$name.so pathtracer_amd/csrc/mipt.hip
