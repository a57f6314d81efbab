#!/bin/bash
# AddressSanitizer + UBSan pass over every kernel (CPU build of the SAME csrc/*.hip sources on tools/hipemu; no GPU sanitizers run on
# this pool): the kernel tests and a tiny model forward / backward with exact-size torch allocations, which ASAN's malloc interposer
# surrounds with red zones -- an out-of-bounds global read or write of a ragged-tile path aborts the run.
#   bash tools/run_asan_emu.sh [pytest -k expression]     -> profiles/r06_asan_emu.txt (written when the run completes)
set -u
cd "$(dirname "$0")/.."
python tools/hipemu/build.py --asan || exit 1
RT=$(python -c "import sys; sys.path.insert(0, 'tools/hipemu'); import build; print(build.asan_runtime())")
FINAL=profiles/r06_asan_emu.txt
OUT=$(mktemp /tmp/asan_emu.XXXXXX)      # (moved over $FINAL only when the run has finished: an interrupted run leaves the last complete report)
K=${1:-}
{
  echo "# $(date -u +%F) tools/run_asan_emu.sh: clang -fsanitize=address,undefined build of bbdm_amd/csrc/*.hip on tools/hipemu"
  echo "# LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0:abort_on_error=1 HIPEMU_ASAN=1 BBDM_TESTS_SERIAL=1"
  echo "# python -m pytest tests/test_emu_kernels_cpu.py tests/test_emu_model_cpu.py tests/test_optim_emu_cpu.py tests/test_egress_emu_cpu.py -q -x ${K:+-k \"$K\"}"
} > $OUT
LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0:abort_on_error=1:verify_asan_link_order=0 \
UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 HIPEMU_ASAN=1 BBDM_TESTS_SERIAL=1 \
  python -m pytest tests/test_emu_kernels_cpu.py tests/test_emu_model_cpu.py tests/test_optim_emu_cpu.py tests/test_egress_emu_cpu.py \
  -q -x -p no:cacheprovider ${K:+-k "$K"} 2>&1 | grep -v "^$" | tail -60 >> $OUT
mv $OUT $FINAL
tail -15 $FINAL
