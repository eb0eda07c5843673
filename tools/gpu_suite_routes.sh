#!/bin/bash
# tools/gpu_suite_routes.sh: the whole -m gpu suite with the library's DEFAULT routes overridden through the environment (the knob
# test covers the combinations on one scene; this runs every test under each alternative).  One test asserts the default itself.
# Round 5: the routes that are new this round first (cross-wave stealing on, with two parameter sets; the hand-over and the thieves'
# bound off; three chunks in flight), then round 4's.  ROUTES=new runs only the former.
NEW=("EZRT_XSTEAL=1" "EZRT_XSTEAL=1 EZRT_XSTEAL_MIN_IDLE=1 EZRT_XSTEAL_STOCK=64 EZRT_XSTEAL_GROUPS=32" "EZRT_HANDOVER=0 EZRT_STEAL_BOUND=0" "EZRT_PIPELINE_DEPTH=3")
OLD=("EZRT_RETREE=0" "EZRT_PRUNE=1 EZRT_REDO_OVERLAP=1" "EZRT_PRUNE=0 EZRT_GEN_PRIMARY=0" "EZRT_SEMI=2 EZRT_ANYHIT=0 EZRT_LAZY_DIR=0 EZRT_TIE_LCA=0" "EZRT_BOUNCE_SCATTER=2 EZRT_PIPELINE_CALLS=2" "EZRT_BOUNCE_SCATTER=0 EZRT_PIPELINE_CALLS=0")
if [ "$ROUTES" = "new" ]; then ALL=("${NEW[@]}"); else ALL=("${NEW[@]}" "${OLD[@]}"); fi
for e in "${ALL[@]}"; do
  echo "== $e"
  env $e timeout 600 python -m pytest tests -q -m gpu --deselect tests/test_gpu_prune.py::test_prune_info_of_the_bunny_scene 2>&1 | grep -E " passed| failed|error" | tail -3
done
