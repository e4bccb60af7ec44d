S=r5_u
timeout 1500 python tests/tools/fuzz_parity.py 1000 9301 > gpurun_out/${S}_fuzz_parity_1000_scenes.txt 2>&1; tail -1 gpurun_out/${S}_fuzz_parity_1000_scenes.txt
timeout 1500 python tests/tools/fuzz_parity.py 500 9302 --queue > gpurun_out/${S}_fuzz_parity_500_scenes_queue.txt 2>&1; tail -1 gpurun_out/${S}_fuzz_parity_500_scenes_queue.txt
timeout 1500 python tests/tools/fuzz_parity.py 300 9303 --queue --spheres > gpurun_out/${S}_fuzz_parity_300_scenes_queue_spheres.txt 2>&1; tail -1 gpurun_out/${S}_fuzz_parity_300_scenes_queue_spheres.txt
timeout 1500 python tests/tools/fuzz_parity.py 200 9304 --kind=merl --merl-tiers --spheres > gpurun_out/${S}_fuzz_parity_200_scenes_measured_brdf_both_tiers.txt 2>&1; tail -1 gpurun_out/${S}_fuzz_parity_200_scenes_measured_brdf_both_tiers.txt
