#!/bin/bash
# Round-4 measurement set (run on a GPU box): the forms of the row kernel, the widths only the brick form takes.
OUT=gpurun_out/r04m; mkdir -p $OUT
B="timeout -k 10 300 python3 bench.py --cpu-frames 0 --no-side --end-to-end 0 --steps 3 --warmup 1"
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1', d['value'], 'frames/s', d['ms_per_step'], 'ms/step  row kernel per window', r['avg_launch_us'], 'us frac', r['frac'], ' isolated', (r.get('isolated') or {}).get('avg_launch_us'), (r.get('isolated') or {}).get('frac'), ' classify', d['kernel_breakdown']['sweep_us'], 'us')"; }
{
for form in sums rows bricks; do for kind in A B; do SAF_WIN_FORM=$form $B --depth-kind $kind 2>/dev/null | line "form=$form depth=$kind 256^3x512 f32:"; done; done
for env in SAF_WIN_SLABS=2 SAF_WIN_SLABS=4 SAF_WIN_W0_SLABS=4; do env $env $B 2>/dev/null | line "form=sums $env depth=A:"; done
for d in 320 640 1280; do $B --grid 128 --dim $d 2>/dev/null | line "default form (bricks: the row kernels do not take this width) 128^3 x $d:"; done
SAF_WIN_FORM=bricks SAF_BRICK_SPLIT=0 $B --grid 128 --dim 640 2>/dev/null | line "bricks without the build kernel 128^3 x 640:"
} > $OUT/bench_forms.txt 2>&1
cat $OUT/bench_forms.txt
