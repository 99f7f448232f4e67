#!/bin/bash
# Everything profiles/r03_* is built from (one gpurun call): bench lines (default incl. CPU baseline + offset sweep, 1 / 2 / 3 / 5 px, config 3),
# config-5 inference line, rocprofv3 kernel stats for default / 1 px / 5 px / config 3, SQ + TCC counter passes of the DCN
# kernels (nf64 L1 shape B=40 at three offset scales, nf128 B=16).
O=gpurun_out/r03_profiles; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python bench.py > $O/bench_default.json 2> $O/bench_default.err
for px in 1 2 3 5; do python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-sweep --offset-px $px > $O/offsets_${px}px.json 2>/dev/null; done
python bench.py --nf 128 --nframes 7 --batch 16 --steps 3 --warmup 1 --no-cpu-baseline --no-sweep > $O/bench_c3.json 2>/dev/null
python bench.py --nf 128 --nframes 7 --batch 16 --steps 3 --warmup 1 --no-cpu-baseline --no-sweep --offset-px 3 > $O/bench_c3_3px.json 2>/dev/null
python tools/infer_clip.py > $O/infer_c5.json 2>/dev/null
for tag in default 3px 5px; do
  flag=""; [ $tag = 3px ] && flag="--offset-px 3"; [ $tag = 5px ] && flag="--offset-px 5"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sweep $flag > $O/prof_$tag.json 2>/dev/null
  cp "$(find $O/prof_$tag -name '*kernel_stats.csv' | head -1)" $O/${tag}_kernel_stats.csv; rm -rf $O/prof_$tag
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c3 -- python bench.py --nf 128 --nframes 7 --batch 16 --steps 2 --warmup 1 --no-cpu-baseline --no-sweep > $O/prof_c3.json 2>/dev/null
cp "$(find $O/prof_c3 -name '*kernel_stats.csv' | head -1)" $O/c3_kernel_stats.csv; rm -rf $O/prof_c3
# counters: nf64 (B=40) at ~0.1 / 1 / 3 / 5 px mean |offset| (std 0.125 / 1.25 / 3.75 / 6.25: the 2 / 5 / 12 / 12 px halos of dcn_bwdin5),
# nf128 (B=16) at 0.125 and 3.75 (2 / 8 px halos, single pass over the 128 output channels)
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"
P2="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU"
P3="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM GRBM_GUI_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_VMEM"
for cfg in "64 40 0.125" "64 40 1.25" "64 40 3.75" "64 40 6.25" "128 16 0.125" "128 16 3.75"; do
  set -- $cfg; C=$1; B=$2; S=$3
  i=0
  for P in "$P1" "$P2" "$P3" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1)); d=$O/pmc_tmp
    rocprofv3 --pmc $P --kernel-trace --output-format csv -d $d -- python tools/dcn_micro.py --iters 2 --B $B --C $C --ostd $S > /dev/null 2>&1
    f=$(find $d -name '*counter_collection.csv' | head -1)
    [ -n "$f" ] && python tools/pmc_summary.py "$f" > $O/pmc_C${C}_std${S}_p$i.txt
    rm -rf $d
  done
done
# conv kernels: counters of the 3x3 64->64 forward on 40 frames (conv_fwd5 wide tile) and of its backward (dgrad + wgrad)
for P in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "FETCH_SIZE" "WRITE_SIZE"; do
  d=$O/pmc_tmp
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d $d -- python tools/conv_micro.py --iters 2 --bwd > /dev/null 2>&1
  f=$(find $d -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python tools/pmc_summary.py "$f" conv >> $O/conv_sq_counters.txt
  rm -rf $d
done
ls $O | head -80
