cd "$(dirname "$0")/.." || exit 1
O=cleverrec_amd/lib/obj
for nb in 4 8 10 20; do
  out=/tmp/crec_cml_$nb; mkdir -p $out
  hipcc --offload-arch=gfx950 -x hip -O3 -fPIC -std=c++17 -ffp-contract=off -Iinclude -DCML_NB=$nb -c cleverrec_amd/csrc/cml.hip -o $out/cml.o || exit 1
  objs=$(ls $O/*.o | grep -v "/cml.o")
  hipcc --offload-arch=gfx950 -shared -o $out/libcrec.so $objs $out/cml.o -L/opt/rocm/lib -lrccl || exit 1
  echo "== NB=$nb"; CREC_LIB=$out/libcrec.so python tools/prof_ncf.py 2>&1 | grep "^CML"
done
