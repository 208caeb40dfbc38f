#!/bin/bash
# Quiet stretch before a wave speculates (FR_DEBUG_SPEC_QUIET iterations; the default is kSpecQuiet in fr_api.hip):
# tools/spec_ab.py per value, each process measuring loop_mode -1 against 5 (no speculation) interleaved.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
for q in ${SPEC_QUIETS:-16 8 32 64 16}; do
  echo "== FR_DEBUG_SPEC_QUIET=$q"
  FR_DEBUG_SPEC_QUIET=$q python3 "$REPO/tools/spec_ab.py" c2 c2f32 gui4k c1 c3 2>&1 | grep "loop_mode -1"
done
