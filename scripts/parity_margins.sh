#!/bin/bash
# GPU suite with the printed parity margins kept: stdout of `pytest -s` (the "grad vs f64 reference" / fit-curve lines)
# and one line per test with the largest relative error any comparison saw (tests/conftest.py, MFVI_MARGINS).  Run on the GPU box;
# copy gpurun_out/margins/* into profiles/r04_parity_margins*.txt afterwards.
cd "$(dirname "$0")/.." || exit 1
OUT=gpurun_out/margins; mkdir -p $OUT
MFVI_MARGINS=$OUT/per_test.tsv python3 -m pytest tests -x -q -m gpu -s -p no:cacheprovider > $OUT/pytest_s.log 2>&1
rc=$?
grep -E "grad vs f64|grad err vs f64|first [0-9]+ its|whole run|bf16 vs f32" $OUT/pytest_s.log > $OUT/printed_lines.txt
tail -3 $OUT/pytest_s.log
exit $rc
