# Ordered kernel sequence of one steady-state training step (rocprofv3 kernel trace; serialised): tools/seq_trace.sh [f32|bf16]
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"
R=$GRAFT_REPO_ROOT
D=${1:-bf16}
O=$R/gpurun_out/seq_$D; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace -d $O/t -o r --output-format csv -- python3 $R/bench.py --legs none --dtype $D --steps 3 --warmup 1 --no-cpu-baseline --no-paint > $O/log.txt 2>&1
python3 $R/tools/kernel_sequence.py $(ls $O/t/*kernel_trace.csv | head -1) > $O/sequence.txt
rm -rf $O/t
wc -l $O/sequence.txt
