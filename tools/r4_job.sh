mkdir -p gpurun_out/r4m
timeout 1200 python tests/tools/fuzz_parity.py 250 4410 --kind=merl --merl-tiers --spheres > gpurun_out/r4_i_fuzz_parity_250_scenes_measured_brdf_both_tiers_spheres.txt 2>&1; tail -2 gpurun_out/r4_i_fuzz_parity_250_scenes_measured_brdf_both_tiers_spheres.txt | cut -c1-300
bash tools/kstats.sh r4_i_c4 --workload c4 | head -8
