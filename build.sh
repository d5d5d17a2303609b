#!/bin/bash
# Builds the GPU library (gfx950), the C++ host program on top of its C ABI, and
# the CPU oracle used by the tests.
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
$HIPCC -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -shared -std=c++17 \
  -Wall -Wno-unused-function -o slam_amd/libmcmcgpu.so slam_amd/csrc/mcmcgpu.hip
# diagnostic twin with per-phase cycle stamps (never used for reported numbers)
$HIPCC -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -shared -std=c++17 -DMG_PHASE_TIMING \
  -Wall -Wno-unused-function -o slam_amd/libmcmcgpu_diag.so slam_amd/csrc/mcmcgpu.hip
# host program: the reference's --mcmc-slam / --multi-mcmc command line over the C ABI
g++ -O2 -std=c++17 -Wall -Wextra -ffp-contract=off -o slam_amd/host/slam_hip slam_amd/host/main.cpp \
  -Lslam_amd -lmcmcgpu -Wl,-rpath,'$ORIGIN/..'
# the same program with the reference's other observation model (range and bearing)
g++ -O2 -std=c++17 -Wall -Wextra -ffp-contract=off -DSLAM_HIP_RANGE_BEARING -o slam_amd/host/slam_hip_rb \
  slam_amd/host/main.cpp -Lslam_amd -lmcmcgpu -Wl,-rpath,'$ORIGIN/..'
make -s -C oracle
