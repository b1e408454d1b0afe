#!/bin/bash
O=gpurun_out
python -m pytest tests/test_kernels_gpu.py tests/test_stages_gpu.py -x -q -k "sinkhorn or matcher" 2>&1 | tail -3 | tee $O/r4_f_pytest.log
(echo "four-wave kernel (nq <= 63):"; python scripts/sinkhorn_one.py; echo "1024-thread kernel (NOPESAC_SINKHORN_NO_W4=1):"; NOPESAC_SINKHORN_NO_W4=1 python scripts/sinkhorn_one.py) 2>&1 | grep -v amdgpu.ids | tee $O/r4_f_sinkhorn.txt
