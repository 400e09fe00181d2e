#!/bin/bash
# Stage the REAL reference (facebookresearch/Pearl, pure Python) under oracle/_ref/ so that it
# travels to the GPU box with the working tree (oracle/_ref/ is git-ignored, like the built .so:
# nothing of the reference enters the history, and nothing under pearl_amd/ ever imports it).
#
# Who uses it (test infrastructure only):
#   tests/test_reference_binding.py   the real pearl.pearl_agent.PearlAgent drives libpearl_amd.so
#                                     on a GPU (observe -> learn -> act)
#   bench.py / bench_algos.py         cpu_baseline.kind == "reference": the reference's own
#                                     PearlAgent.learn timed on the host cores (child process with
#                                     no visible GPU: pearl/utils/device.py:48-59)
# Run by __graft_entry__.build() whenever /root/reference is present (the build container).
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
REF="${PEARL_REFERENCE:-/root/reference}"
if [ ! -d "$REF/pearl" ]; then
  echo "stage_ref: no reference at $REF (GPU box?) - keeping whatever oracle/_ref holds" >&2
  exit 0
fi
mkdir -p "$HERE/_ref"
rm -rf "$HERE/_ref/pearl"
# the package only: no tests, tutorials or docs; bytecode caches stay behind
(cd "$REF" && find pearl -name '__pycache__' -prune -o -type f -name '*.py' -print0 \
  | xargs -0 -I{} cp --parents {} "$HERE/_ref/")
[ -f "$REF/LICENSE" ] && cp "$REF/LICENSE" "$HERE/_ref/LICENSE"
echo "staged from $REF: $(find "$HERE/_ref/pearl" -name '*.py' | wc -l) files" > "$HERE/_ref/STAGED"
cat "$HERE/_ref/STAGED"
