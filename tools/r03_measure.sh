#!/bin/bash
# Round-3 measurement set (run on a GPU box): writes small JSON / text files under gpurun_out/r03/.
OUT=gpurun_out/r03; mkdir -p $OUT
B="timeout -k 10 300 python3 bench.py --cpu-frames 0 --no-side --steps 2 --warmup 1"
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1', d['value'], 'frames/s', d['ms_per_step'], 'ms/step  row kernel', r['avg_launch_us'], 'us frac', r['frac'], ' isolated', (r.get('isolated') or {}).get('avg_launch_us'), ' classify', d['kernel_breakdown']['sweep_us'], 'us')"; }
{
for form in rows bricks; do for kind in A B; do SAF_WIN_FORM=$form $B --depth-kind $kind 2>/dev/null | line "form=$form depth=$kind 256^3x512 f32:"; done; done
SAF_WIN_FORM=bricks SAF_BRICK_SPLIT=0 $B 2>/dev/null | line "form=bricks (no build kernel) depth=A:"
for g in 127,104,116 118,115,113 128; do for form in rows bricks; do SAF_WIN_FORM=$form $B --grid $g 2>/dev/null | line "form=$form grid=$g x512:"; done; done
for d in 320 640 1280; do $B --grid 128 --dim $d 2>/dev/null | line "default form (bricks: the row kernel does not take this width) 128^3 x $d:"; SAF_WINDOW=0 $B --grid 128 --dim $d 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   per-frame pipeline (round 2 behaviour) 128^3 x $d:', d['value'], 'frames/s')"; done
} > $OUT/bench_forms.txt 2>&1
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/lds_atomic_bench.hip -o /tmp/lds_atomic_bench 2>/dev/null && timeout -k 5 120 /tmp/lds_atomic_bench > $OUT/lds_atomic_bench.log
SAF_BENCH_ONE_DEVICE=1 timeout -k 10 600 python3 bench.py --gpus 2 --backend gloo --steps 1 --warmup 1 --grid 128 --frames 128 --unique-frames 128 --cpu-frames 0 > $OUT/rehearsal_gpus2_gloo_one_device.json 2> $OUT/rehearsal.err
SAF_WIN_FORM=bricks bash tools/pmc_brick.sh gpurun_out/r03/pmcbrick --no-side > $OUT/pmc_brick.log 2>&1
cp gpurun_out/r03/pmcbrick/brick_pmc.json $OUT/brick_pmc.json; rm -rf gpurun_out/r03/pmcbrick
SAF_WIN_FORM=rows bash tools/pmc_brick.sh gpurun_out/r03/pmcrows --no-side > $OUT/pmc_rows.log 2>&1
cp gpurun_out/r03/pmcrows/brick_pmc.json $OUT/rows_pmc.json; rm -rf gpurun_out/r03/pmcrows
timeout -k 10 600 python3 bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
echo done
