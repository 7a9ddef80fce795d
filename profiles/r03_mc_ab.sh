# Same-box A/B of gf2_mc_run at n = 4096: the working tree's library against scratch_ab/<name>.so, alternating, three rounds.
#   bash profiles/r03_mc_ab.sh <name>
cp quantum_css_codes_amd/libgf2hip.so /tmp/new.so
for i in 1 2 3; do for w in new $1; do
  if [ $w = new ]; then cp /tmp/new.so quantum_css_codes_amd/libgf2hip.so; else cp scratch_ab/$w.so quantum_css_codes_amd/libgf2hip.so; fi
  echo "$w $(python3 profiles/time_mc.py | tail -n 1)"
done; done
cp /tmp/new.so quantum_css_codes_amd/libgf2hip.so
