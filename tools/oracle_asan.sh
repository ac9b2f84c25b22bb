#!/bin/bash
# CPU suite on the AddressSanitizer + UBSan build of the oracle (oracle/Makefile: libps_oracle_asan.so).
# CPU only: GPU sanitizers are not available on this pool.
set -e
cd "$(dirname "$0")/.."
make -C oracle libps_oracle_asan.so
ASAN=$(gcc -print-file-name=libasan.so)
PORESEQ_ORACLE_SO=$PWD/oracle/libps_oracle_asan.so LD_PRELOAD=$ASAN ASAN_OPTIONS=detect_leaks=0 \
  python -m pytest tests/test_oracle.py tests/test_dist.py -x -q -m "not gpu" "$@"
