#!/bin/bash
# build/ab/libga_<name>.so = the product with extra -D flags (same-box A/B runs: bench.py --lib build/ab/libga_<name>.so)
set -e
name=$1; shift
cd "$(dirname "$0")/.."
mkdir -p build/ab
[ -f build/ga_host.o ] || python -c "import __graft_entry__ as e; e.build_product(True)"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -DGA_WAVES_EU=6 "$@" -c graphaligner_amd/csrc/ga_device.hip -o build/ab/ga_device_$name.o 2> build/ab/$name.log
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -pthread -o build/ab/libga_$name.so build/ga_host.o build/ga_vgio.o build/ab/ga_device_$name.o -lz
rm -f build/ab/ga_device_$name.o
grep -c "spill\|scratch" build/ab/$name.log || true
