# The phase clocks of the sweep kernels (gf2_elim.hip, GF2_SWEEP_DIAG): a variant library with the stamps compiled in, then one shape.
#   here:        bash profiles/build_variant.sh diag - -DGF2_SWEEP_DIAG=1
#   on the box:  bash profiles/r05_diag.sh "2048 4096 256 4"      (m n batch K; GF2_RREF_DIAG_WG=<file> logs one pass' workgroups)
root=$(pwd)
cp quantum_css_codes_amd/libgf2hip.so /tmp/base.so && cp scratch_ab/diag.so quantum_css_codes_amd/libgf2hip.so
GF2_RREF_DIAG=1 python3 profiles/r05_rref_one.py $1
cp /tmp/base.so quantum_css_codes_amd/libgf2hip.so
