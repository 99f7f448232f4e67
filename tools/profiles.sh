#!/bin/bash
# Everything profiles/<round>_* is built from (one gpurun call): tools/profiles.sh r06 -- the default driver line, config 3, config 5, the forced one-rank
# all-reduce line, rocprofv3 kernel stats of the default and the config-3 step, and PMC passes (HBM traffic + SQ counters; separate --pmc passes with
# --kernel-trace only) of the DCN forward at nf64 / nf128 and of the DCN backward pair, all at offset std 1.25 px.  tools/profiles_collect.py <round>
# turns the output into profiles/<round>_*.
R=${1:-r06}
O=gpurun_out/${R}_profiles; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --config 3 --steps 3 --warmup 1 --no-cpu-baseline --no-sweep --no-extra > $O/bench_c3.json 2>/dev/null
python bench.py --config 5 > $O/infer_c5.json 2>/dev/null
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-sweep --no-extra --force-allreduce > $O/bench_force_allreduce.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_default -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sweep --no-extra > $O/prof_default.json 2>/dev/null
cp "$(find $O/prof_default -name '*kernel_stats.csv' | head -1)" $O/default_kernel_stats.csv; rm -rf $O/prof_default
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c3 -- python bench.py --config 3 --steps 2 --warmup 1 --no-cpu-baseline --no-sweep --no-extra > $O/prof_c3.json 2>/dev/null
cp "$(find $O/prof_c3 -name '*kernel_stats.csv' | head -1)" $O/c3_kernel_stats.csv; rm -rf $O/prof_c3
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"
P2="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU"
P3="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"
for cfg in "fwd64 64 40 --fwd-only" "fwd128 128 16 --fwd-only" "bwd64 64 40" "bwd128 128 16"; do
  set -- $cfg
  i=0
  for P in "$P1" "$P2" "$P3" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1)); d=$O/pmc_tmp
    rocprofv3 --pmc $P --kernel-trace --output-format csv -d $d -- python tools/dcn_micro.py --iters 2 --B $3 --C $2 --ostd 1.25 $4 > /dev/null 2>&1
    f=$(find $d -name '*counter_collection.csv' | head -1)
    [ -n "$f" ] && python tools/pmc_summary.py "$f" > $O/pmc_$1_p$i.txt
    rm -rf $d
  done
done
ls $O
