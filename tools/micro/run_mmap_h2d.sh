#!/bin/bash
# file (tmpfs) <-> HBM paths, measured: mmap + hipHostRegister + DMA against pread into page-locked blocks; and the way out: D2H into page-locked blocks + pwrite on T threads
# against memcpy into an mmap'ed, pre-sized output.  usage (on the box): bash tools/micro/run_mmap_h2d.sh
set -u
cd ${GRAFT_REPO_ROOT:-.}
D=/dev/shm/mm; mkdir -p $D
./tools/fqgen --profile 1 --reads 5600000 --seed 3 -o $D/r1.fq -O $D/r2.fq
/opt/rocm/bin/hipcc -O2 -o /tmp/mmap_h2d tools/micro/mmap_h2d.cpp -lpthread && /tmp/mmap_h2d $D/r1.fq 256
/opt/rocm/bin/hipcc -O2 -o /tmp/d2h_out tools/micro/d2h_out.cpp -lpthread && /tmp/d2h_out $D/out.bin 4000
rm -rf $D
