O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_zipenhancer.py -m gpu -x -q -s > $O/r05_k_tests.txt 2>&1; echo "tests rc $?"; tail -2 $O/r05_k_tests.txt; grep -a "bf16" $O/r05_k_tests.txt | cut -c1-400
for CH in 1 0 1 0; do ADE_ZIP_CHAIN=$CH timeout 600 python bench.py --workload zipenhancer --dtype bf16 --cpu-seconds 0 --host-steps 0 --no-deviation 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('chain $CH', d['ms_per_step'])"; done
python - <<'PY'
import os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from audio_denoiser_onnx_amd import zipenhancer as zp
from audio_denoiser_onnx_amd.weights import pack_blob
from audio_denoiser_onnx_amd.session import InferenceSession
from audio_denoiser_onnx_amd.synth import synth_batch
t = zp.synthetic_tensors(zp.ZipConfig()); blob = pack_blob(t); x = synth_batch(8, 16000)
outs = []
for ch in ("1", "0"):
    os.environ["ADE_ZIP_CHAIN"] = ch
    with InferenceSession(weights=blob, metadata=zp.metadata(16000, gemm_dtype="bf16")) as s:
        outs.append(s.process(x, want_f32=True))
print("chain == two-kernel form bit for bit:", np.array_equal(outs[0][0], outs[1][0]), np.array_equal(outs[0][1], outs[1][1]))
PY
