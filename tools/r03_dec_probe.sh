#!/bin/bash
# Round 3, decoder: GPU parity tests of the decoder, then decode-only passes at several batch sizes with kernel statistics.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests/test_gpu_decode.py tests/test_color_space.py tests/test_records_api.py tests/test_gpu_host_cli.py -x -q -m gpu 2>&1 | tail -8 ) > gpurun_out/r03_dec_pytest.txt
cat gpurun_out/r03_dec_pytest.txt
rocprofv3 --list-avail 2>/dev/null | grep -i -E "UTCL|TLB|TCC_HIT|TCC_MISS|TCC_REQ|TCC_EA_RDREQ|TCC_EA_WRREQ|MALL|TCP_TCC|LATENCY|TCC_TAG_STALL|TCC_BUSY" | cut -c1-160 | sort | uniq | head -80 > gpurun_out/r03_counters.txt
bash tools/r03_dec_probe2.sh
