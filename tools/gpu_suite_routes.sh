#!/bin/bash
# tools/gpu_suite_routes.sh: the whole -m gpu suite with the library's DEFAULT routes overridden through the environment (the knob
# test covers the combinations on one scene; this runs every test under each alternative).  One test asserts the default itself.
# Round 6: the routes that exist after the pruning -- the stack ring at 4 / 8 / 32 rows (4: every deep ray spills), the hand-over and the
# thieves' bound off, the re-tree off, pruning modes 1 / 0 (their exact slot-order rows, no ring), raygen_kernel
# instead of in-launch generation, every exact tie and zero-component ray through the redo list or all of them in the wide kernel,
# consecutive slots instead of the scattered draw, chunks never pipelined, small chunks.  ROUTES=new runs only the ring routes.
NEW=("EZRT_STACK_CAP=4" "EZRT_STACK_CAP=8 EZRT_HANDOVER=0 EZRT_STEAL_BOUND=0" "EZRT_STACK_CAP=32 EZRT_CHUNK_LOG2=20")
OLD=("EZRT_RETREE=0" "EZRT_PRUNE=1 EZRT_PRUNE_MIS=1" "EZRT_PRUNE=0 EZRT_GEN_PRIMARY=0" "EZRT_SEMI=2 EZRT_ANYHIT=0 EZRT_LAZY_DIR=0 EZRT_TIE_LCA=0" "EZRT_SEMI=0 EZRT_TIE_LCA=0 EZRT_STEAL=0" "EZRT_BOUNCE_SCATTER=0 EZRT_PIPELINE_CALLS=0" "EZRT_WIDE4=0")
if [ "$ROUTES" = "new" ]; then ALL=("${NEW[@]}"); else ALL=("${NEW[@]}" "${OLD[@]}"); fi
for e in "${ALL[@]}"; do
  echo "== $e"
  env $e timeout 900 python -m pytest tests -q -m gpu --deselect tests/test_gpu_prune.py::test_prune_info_of_the_bunny_scene 2>&1 | grep -E " passed| failed|error" | tail -3
done
