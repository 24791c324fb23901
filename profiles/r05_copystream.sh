# host-fed paths with the copy stream picked by bsms_streams_overlap: fresh variable meshes, Trainer.iter from host batches
cd "$(dirname "$0")/.."; mkdir -p gpurun_out/r05cs
timeout 600 python -m pytest tests/test_hip_host_builder.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2
timeout 300 python profiles/fresh_mesh.py 8 2>&1 | grep -v amdgpu | tail -5
timeout 300 python profiles/trainer_loop.py airfoil 8 2>&1 | grep -v amdgpu | tail -3
timeout 300 python profiles/trainer_loop.py cylinder 8 2>&1 | grep -v amdgpu | tail -3
