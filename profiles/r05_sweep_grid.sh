# x256 under K x groups x pass threads (no profiler):  bash profiles/r05_sweep_grid.sh
for k in 4 2; do for g in 1 2 4; do for th in 1024 512; do
python3 profiles/r05_rref_one.py 2048 4096 256 $k $g -1 $th | tail -1
done; done; done
python3 profiles/r05_rref_one.py 2048 4096 1 4 1 -1 512 | tail -1
python3 profiles/r05_rref_one.py 2048 4096 1 2 1 -1 512 | tail -1
