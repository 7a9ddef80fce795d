# Same-box A/B of gf2_rref_batch_dev: the working tree's library against scratch_ab/<name>.so, alternating.
#   bash profiles/r03_rref_ab.sh <name>
cp quantum_css_codes_amd/libgf2hip.so /tmp/new.so
for i in 1 2 3; do for w in new $1; do
  if [ $w = new ]; then cp /tmp/new.so quantum_css_codes_amd/libgf2hip.so; else cp scratch_ab/$w.so quantum_css_codes_amd/libgf2hip.so; fi
  for shape in "2048 4096 256" "2048 4096 1" "1000 3000 64" "4096 8192 16"; do
    echo "$w $(python3 profiles/time_rref.py $shape | tail -n 1)"
  done
done; done
cp /tmp/new.so quantum_css_codes_amd/libgf2hip.so
