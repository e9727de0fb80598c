#!/bin/bash
# A/B of the video training legs: clip layout with deferred weight gradients (UNCL_CLIP_WGRAD=1, default) against the per-frame form
out=gpurun_out/${1:-ab_clip}; mkdir -p $out
for v in 1 0 1 0; do for leg in train_video_step train_video_step_b8; do
  UNCL_CLIP_WGRAD=$v timeout 300 python bench.py --leg $leg --no-cpu > $out/${leg}_$v.json 2> $out/${leg}_$v.err
  echo "$leg clip=$v rc=$?"
  python - $out/${leg}_$v.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print({k: d[k] for k in d if not isinstance(d[k], (dict, list))})
except Exception as e:
    print("ERR", e)
PY
done; done
