#!/bin/bash
# Development tool (GPU box): the chunk roots' tree finished by the last chunk workgroup of the same launch (SMI_MERKLE_TICKET=1)
# against the two-launch default: parity first (Merkle / FRI / prove tests under the knob), then three alternating rounds.
set -e
export TMPDIR=/tmp
SMI_MERKLE_TICKET=1 timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_pipeline.py tests/test_gpu_verify.py tests/test_gpu_mgpu.py -x -q > gpurun_out/r03_ticket_tests.log 2>&1 || { tail -30 gpurun_out/r03_ticket_tests.log; exit 1; }
tail -2 gpurun_out/r03_ticket_tests.log
{ for i in 1 2 3; do python3 tools/prove_time.py 22 two-launches; SMI_MERKLE_TICKET=1 python3 tools/prove_time.py 22 ticket; done
  for i in 1 2 3; do python3 tools/prove_time.py 20 two-launches; SMI_MERKLE_TICKET=1 python3 tools/prove_time.py 20 ticket; done; } 2>/dev/null > gpurun_out/r03_ticket_ab.log
cat gpurun_out/r03_ticket_ab.log
SEED=4242 SMI_MERKLE_TICKET=1 timeout -k 10 300 python3 tools/stress.py 200 > gpurun_out/stress_r03_ticket.log 2>&1; tail -2 gpurun_out/stress_r03_ticket.log
