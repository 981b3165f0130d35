python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_classes.py -x -q -k "given_P or batched" 2>&1 | tail -n 2
for rep in 1 2; do
for v in base rows_b2_u8 rows_sc1 rows_b4_u8; do echo $v; if [ $v = base ]; then python tools/step_probe.py 128,16384 256,16384 2>&1 | grep "d="; else MFG_HIP_LIB=discrete_mean_field_game_amd/csrc/variants/lib$v.so python tools/step_probe.py 128,16384 256,16384 2>&1 | grep "d="; fi; done
done
