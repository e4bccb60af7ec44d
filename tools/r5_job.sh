bash tools/closing_record.sh r5_k 3716b49 > gpurun_out/r5_k_closing_record.log 2>&1
tail -40 gpurun_out/r5_k_closing_record.log
