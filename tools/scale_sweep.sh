#!/bin/bash
# Scaling sweep on one node: bench.py at 1, 2, 4, 8 GPUs x {1024, 512} (north_star: iterations/s at both sizes at 1, 2, 4 and 8 GPUs).
# One JSON line per run goes to $OUT (default gpurun_out/scale_sweep.jsonl).  N = 1 runs in-process; N > 1 goes through torchrun
# exactly as the driver launches it (one rank per GPU over RCCL, 127.0.0.1 rendezvous).  With fewer GPUs than asked the run is
# skipped and recorded as such (bench.py --allow_fewer is the degrade-and-report form for a by-hand run).
# usage: tools/scale_sweep.sh [steps] [warmup]
set -u
cd "$(dirname "$0")/.."
STEPS=${1:-100}
WARMUP=${2:-5}
OUT=${OUT:-gpurun_out/scale_sweep.jsonl}
mkdir -p "$(dirname "$OUT")"
: > "$OUT"
HAVE=$(python -c 'import torch; print(torch.cuda.device_count())')
export HSA_ENABLE_IPC_MODE_LEGACY=${HSA_ENABLE_IPC_MODE_LEGACY:-0}
PORT=29600
for SIZE in 1024 512; do
  for N in 1 2 4 8; do
    if [ "$N" -gt "$HAVE" ]; then
      echo "{\"skipped\": true, \"n_gpus\": $N, \"image_size\": $SIZE, \"visible\": $HAVE}" >> "$OUT"
      continue
    fi
    FLAGS="--gpus $N --steps $STEPS --warmup $WARMUP --size $SIZE --no_cpu_baseline --no_exact_split"
    [ "$SIZE" != 1024 ] && FLAGS="$FLAGS --no_extra_sizes"
    if [ "$N" -eq 1 ]; then
      python bench.py $FLAGS | grep '^{' >> "$OUT"
    else
      PORT=$((PORT + 1))
      python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port "$PORT" \
        bench.py $FLAGS | grep '^{' >> "$OUT"
    fi
  done
done
python - "$OUT" <<'PY'
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1]) if l.strip()]
base = {}
for r in rows:
    if r.get("skipped"):
        print(f"{r['image_size']:>5} px  N={r['n_gpus']}: skipped ({r['visible']} GPU(s) visible)")
        continue
    size = r["config"]["image_size"]
    base.setdefault(size, r["value"] / r["n_gpus"] if r["n_gpus"] == 1 else None)
    eff = f"  {r['value'] / (r['n_gpus'] * base[size]):.3f} of linear" if base.get(size) else ""
    print(f"{size:>5} px  N={r['n_gpus']}: {r['value']:.1f} it/s{eff}")
PY
