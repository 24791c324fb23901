mkdir -p gpurun_out/r05v; cd /root/repo
for p in f32 bf16 bf16_nodes; do python profiles/b1_rates.py airfoil 1 $p 2>&1 | tail -1; done > gpurun_out/r05v/b1.txt
