#!/bin/bash
# A/B of crec_sampler.cpp revisions on ONE box: builds libcrec.so with the sampler of each given commit under /tmp and runs
# tools/prof_epoch.py with it (GPU box; the tree must be a git checkout -- on gpurun it is not, so pass files instead:
# usage: tools/sampler_ab.sh path/to/old_sampler.cpp cleverrec_amd/csrc/crec_sampler.cpp
cd "$(dirname "$0")/.." || exit 1
O=cleverrec_amd/lib/obj
n=0
for src in "$@"; do
  n=$((n+1)); out=/tmp/crec_samp_$n; mkdir -p $out
  cp "$src" cleverrec_amd/csrc/_ab_sampler.cpp
  hipcc --offload-arch=gfx950 -x hip -O3 -fPIC -std=c++17 -fno-math-errno -pthread -Iinclude -c cleverrec_amd/csrc/_ab_sampler.cpp -o $out/crec_sampler.o || exit 1
  rm -f cleverrec_amd/csrc/_ab_sampler.cpp
  objs=$(ls $O/*.o | grep -v "/crec_sampler.o")
  hipcc --offload-arch=gfx950 -shared -o $out/libcrec.so $objs $out/crec_sampler.o -L/opt/rocm/lib -lrccl -pthread || exit 1
  echo "== $src"; CREC_LIB=$out/libcrec.so python tools/prof_epoch.py 2>&1 | grep "^prefetch"
done
