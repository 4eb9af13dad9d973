timeout 900 python -m pytest tests/test_gpu_stage.py -q 2>&1 | tail -6
for w in form1:conv7 form8:conv7; do echo "== $w"; SMELTER_LIB_PATH=build/libsmelter_hip_stagetl.so timeout 120 python tools/stage_tl.py $w 2>&1 | grep -E "^[1] |kernel:"; done
echo "== stage4 8w"; SMELTER_LIB_PATH=build/libsmelter_hip_stagetl.so timeout 120 python tools/stage_tl.py stage4 2>&1 | grep -E "^[0-9] |kernel:|status"
echo "== stage4 4w"; SMELTER_LIB_PATH=build/libsmelter_hip_stagetl.so timeout 120 python tools/stage_tl.py stage4 4w 2>&1 | grep -E "^[0-9] |kernel:|status"
