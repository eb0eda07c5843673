#!/bin/bash
# tools/gpu_suite_routes.sh: the whole -m gpu suite with the library's DEFAULT routes overridden through the environment (the knob
# test covers the combinations on one scene; this runs every test under each alternative).  One test asserts the default itself.
for e in "EZRT_RETREE=0" "EZRT_PRUNE=1 EZRT_REDO_OVERLAP=1" "EZRT_PRUNE=0 EZRT_GEN_PRIMARY=0" "EZRT_SEMI=2 EZRT_ANYHIT=0 EZRT_LAZY_DIR=0 EZRT_TIE_LCA=0" "EZRT_BOUNCE_SCATTER=2 EZRT_PIPELINE_CALLS=2" "EZRT_BOUNCE_SCATTER=0 EZRT_PIPELINE_CALLS=0"; do
  echo "== $e"
  env $e timeout 900 python -m pytest tests -q -m gpu --deselect tests/test_gpu_prune.py::test_prune_info_of_the_bunny_scene 2>&1 | tail -2
done
