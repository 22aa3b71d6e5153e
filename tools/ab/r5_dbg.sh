set -u
cd $GRAFT_REPO_ROOT
g++ -std=c++17 -O1 tests/cpp/integration_test.cpp -o /tmp/integration_test -Linfercam_onnx_amd -lufacehip -Wl,-rpath,$PWD/infercam_onnx_amd
python3 -c "
from infercam_onnx_amd import synth; import numpy as np
np.asarray(synth.synthetic_weights(), np.float32).tofile('/tmp/w.f32')"
timeout -k 5 60 stdbuf -o0 /tmp/integration_test tests/golden/test_pics /tmp/w.f32 > gpurun_out/integ.txt 2>&1; echo rc=$?
cat gpurun_out/integ.txt
