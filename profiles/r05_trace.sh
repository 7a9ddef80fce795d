# kernel traces of one RREF shape under the internal options: bash profiles/r05_trace.sh "<m n batch K groups>" tag
root=$(pwd); out=$root/gpurun_out/r05; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
args="$1"; tag="$2"
rocprofv3 --kernel-trace --output-format csv -d $out/tr_$tag -- python3 $root/profiles/r05_rref_one.py $args > $out/tr_$tag.log 2>&1 || exit 1
cd $root
python3 profiles/summarize.py $(find $out/tr_$tag -name '*kernel_trace.csv') > $out/tr_$tag.md
cat $out/tr_$tag.log | tail -1; cat $out/tr_$tag.md
