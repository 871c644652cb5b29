#!/bin/bash
# round 5, on the GPU box: what the classification and the row kernel share when they run side by side.
#   1. variants of the classification's lane -> voxel mapping (SAF_CLS_CUBE) and the gather-doubling ablation (SAF_CLS_ABL=1:
#      same results, twice the depth gathers), each timed with the default job, overlap on and off;
#   2. the counter set of profiles/r04/rows_pmc.json per dispatch with SAF_WIN_OVERLAP=1 and =0 (tools/pmc_cores.py
#      condenses it and says whether the dispatches overlapped in time under the profiler at all).
# Usage: bash tools/r05_cores.sh <outdir>
OUT=${1:-gpurun_out/r05a}
mkdir -p $OUT
C=spatially_aware_ai_amd/csrc
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -Wno-unused-function"
OTHERS=$(ls $C/*.o | grep -v saf_window.o | tr "\n" " ")
B="python3 bench.py --cpu-frames 0 --no-side --end-to-end 0 --no-pmc"
line() { python3 -c "
import sys,json
t=sys.stdin.read().strip().splitlines()
try:
    d=json.loads(t[-1]); r=d['roofline']
    print('$1', d['value'], 'frames/s', d['ms_per_step'], 'ms; rows/window', r['avg_launch_us'], 'frac', r['frac'], 'alone', (r.get('isolated') or {}).get('avg_launch_us'), (r.get('isolated') or {}).get('frac'), 'classify', d['kernel_breakdown']['sweep_us'])
except Exception as e:
    print('$1 FAILED', e, t[-3:])"; }
for v in ${VARIANTS:-base CUBE=0 CUBE=0+ABL=1 CUBE=1+ABL=1}; do
  [ "$v" = skip ] && continue
  tag=$(echo $v | tr '+=' '__')
  def=""; [ "$v" != base ] && def=$(echo $v | sed 's/^/-DSAF_CLS_/; s/+/ -DSAF_CLS_/g')
  /opt/rocm/bin/hipcc $FLAGS $def -c $C/saf_window.hip -o /tmp/win_$tag.o && \
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libsaf_$tag.so /tmp/win_$tag.o $OTHERS || { echo "$tag BUILD FAILED"; continue; }
  for k in ${KINDS:-A B}; do
    SAF_LIB_PATH=/tmp/libsaf_$tag.so timeout -k 10 200 $B --steps 8 --warmup 3 --depth-kind $k 2>$OUT/${tag}_$k.err | tee $OUT/${tag}_$k.json | line "$tag depth=$k"
  done
done 2>&1 | tee $OUT/variants.txt
[ -n "$NO_PMC" ] && exit 0
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 -L > $OUT/counters.txt 2>&1 || true
P="python3 bench.py --cpu-frames 0 --frames 512 --steps 1 --warmup 0 --no-profile-events --no-side --end-to-end 0 --no-pmc"
run() { ov=$1; name=$2; shift 2; SAF_WIN_OVERLAP=$ov timeout -k 5 150 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/pmc_ov$ov/$name -- $P > $OUT/pmc_ov$ov/$name.json 2> $OUT/pmc_ov$ov/$name.err || echo "FAILED ov$ov $name"; }
for ov in 1 0; do
  mkdir -p $OUT/pmc_ov$ov
  run $ov a GRBM_GUI_ACTIVE TA_TA_BUSY_sum &&
  run $ov b TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum &&
  run $ov c TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum &&
  run $ov d TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum &&
  run $ov e SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS &&
  run $ov f SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES &&
  run $ov g SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_MISSES &&
  run $ov h TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum &&
  run $ov i TA_BUFFER_WAVEFRONTS_sum TA_BUFFER_READ_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum
done
# keep the csv files the summary needs, drop the rest (the merge back is limited to 64 MiB)
python3 tools/pmc_cores.py $OUT > $OUT/coresident_pmc.txt 2>&1; tail -40 $OUT/coresident_pmc.txt
find $OUT -name "*.csv" -size +2M -delete
