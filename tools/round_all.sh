bash tools/measure_round.sh $1
bash tools/configs_round.sh gpurun_out/$1/configs > /dev/null 2>&1; cat gpurun_out/$1/configs/other_configs.txt | cut -c1-300
