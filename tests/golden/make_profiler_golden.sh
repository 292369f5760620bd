#!/bin/bash
# Regenerates the known-answer outputs of the reference's analytic profiler programs (the only reference
# code that compiles).  Run from the repo root in a container that has /root/reference.
set -e
make -C oracle -s ref
oracle/_ref/profiler 2>/dev/null > tests/golden/profiler_ref.txt
oracle/_ref/profiler_performance_model 2>/dev/null > tests/golden/profiler_performance_model_ref.txt
