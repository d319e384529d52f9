#!/bin/bash
# file on tmpfs -> HBM: mmap + hipHostRegister (no CPU copy) against pread into page-locked blocks.  usage (on the box): bash tools/micro/run_mmap_h2d.sh
cd $GRAFT_REPO_ROOT; mkdir -p /dev/shm/mm
./tools/fqgen --profile 1 --reads 5600000 --seed 3 -o /dev/shm/mm/r1.fq -O /dev/shm/mm/r2.fq
/opt/rocm/bin/hipcc -O2 -o /tmp/mmap_h2d tools/micro/mmap_h2d.cpp -lpthread 2>&1 | tail -2
/tmp/mmap_h2d /dev/shm/mm/r1.fq 64
/tmp/mmap_h2d /dev/shm/mm/r1.fq 16 | head -4
rm -rf /dev/shm/mm
