#!/bin/bash
# Everything profiles/r04_* is built from (one gpurun call): the default driver line (1 px offsets, parity, extras), the 3 px line, rocprofv3
# kernel stats of the default and the config-3 step, HBM traffic + SQ counters of the DCN forward (dcn_fwd3 and dcn_fwd4) at offset std 1.25 px.
O=gpurun_out/r04_profiles; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-sweep --no-extra --offset-px 3 > $O/offsets_3px.json 2>/dev/null
python bench.py --config 3 --steps 3 --warmup 1 --no-cpu-baseline --no-sweep --no-extra > $O/bench_c3.json 2>/dev/null
python bench.py --config 5 > $O/infer_c5.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_default -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sweep --no-extra > $O/prof_default.json 2>/dev/null
cp "$(find $O/prof_default -name '*kernel_stats.csv' | head -1)" $O/default_kernel_stats.csv; rm -rf $O/prof_default
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c3 -- python bench.py --config 3 --steps 2 --warmup 1 --no-cpu-baseline --no-sweep --no-extra > $O/prof_c3.json 2>/dev/null
cp "$(find $O/prof_c3 -name '*kernel_stats.csv' | head -1)" $O/c3_kernel_stats.csv; rm -rf $O/prof_c3
RVSR_GEMM=bf16 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bf16 -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sweep --no-extra > $O/prof_bf16.json 2>/dev/null
cp "$(find $O/prof_bf16 -name '*kernel_stats.csv' | head -1)" $O/bf16_mode_kernel_stats.csv; rm -rf $O/prof_bf16
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"
P2="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU"
P3="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"
for gen in 3 4; do
  i=0
  for P in "$P1" "$P2" "$P3" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1)); d=$O/pmc_tmp
    RVSR_DCN_FWD=$gen rocprofv3 --pmc $P --kernel-trace --output-format csv -d $d -- python tools/dcn_micro.py --iters 2 --B 40 --C 64 --ostd 1.25 --fwd-only > /dev/null 2>&1
    f=$(find $d -name '*counter_collection.csv' | head -1)
    [ -n "$f" ] && python tools/pmc_summary.py "$f" > $O/pmc_fwd${gen}_p$i.txt
    rm -rf $d
  done
done
ls $O
