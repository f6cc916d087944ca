#!/bin/bash
# Build alternative libtripled_hip.so variants of the photometric backward kernel (same ABI) for A/B timing.
# usage: tools/build_variants.sh name "extra hipcc flags" [name "flags" ...]
set -e
PKG=$(cd "$(dirname "$0")/.." && pwd)/tripled-exploring-depth-estimation-with-self-supervised-representation-learning_amd
cd $PKG/csrc
make -s
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -I$PKG/../include -Wall -Wno-unused-function"
while [ $# -gt 1 ]; do
  name=$1; extra=$2; shift 2
  out=$PKG/lib/variants/$name; mkdir -p $out
  /opt/rocm/bin/hipcc $FLAGS $extra -c td_photo_bwd.hip -o $out/td_photo_bwd.o
  /opt/rocm/bin/hipcc $FLAGS $extra -c td_photo_fwd.hip -o $out/td_photo_fwd.o
  objs=$(ls $PKG/lib/obj/*.o | grep -v "td_photo_bwd.o\|td_photo_fwd.o")
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $out/libtripled_hip.so $objs $out/td_photo_bwd.o $out/td_photo_fwd.o
  echo built $out
done
