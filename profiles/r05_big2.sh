# Streamed sweeps (more than 4096 rows): parity on tall shapes, then the launcher's development switches on a few shapes.
#   GF2_OPT_RREF_STREAM_VARIANT (6th argument of r05_rref_one.py): bit 0 = the big launch waits for the side stream to start,
#   bits 1-2 = 1: panels always on the side stream, 2: the big launch always; (R + 1) << 8 = R CUs left to the panels (R / 8 per XCD).
#   7th argument: context flags, 0x10000 = GF2_F_RREF_NO_LOOKAHEAD.
root=$(pwd); out=$root/gpurun_out/r05; mkdir -p $out
python3 profiles/r05_rref_dev.py tall > $out/tall.log 2>&1 || { tail -20 $out/tall.log; exit 1; }
tail -1 $out/tall.log
{
  for v in -1 1 2 4; do python3 profiles/r05_rref_one.py 32768 65536 1 -1 -1 $v || exit 1; done
  for r in 8 16 24 32 40 48 56; do python3 profiles/r05_rref_one.py 32768 65536 1 -1 -1 $(( (r + 1) * 256 )) || exit 1; done
  python3 profiles/r05_rref_one.py 32768 65536 1 -1 -1 -1 0x10000
  for v in -1 2 4; do
    python3 profiles/r05_rref_one.py 8192 16384 4 -1 -1 $v; python3 profiles/r05_rref_one.py 16384 32768 1 -1 -1 $v; python3 profiles/r05_rref_one.py 8192 8192 1 -1 -1 $v
  done
} > $out/big_variants.log 2>&1
cat $out/big_variants.log
