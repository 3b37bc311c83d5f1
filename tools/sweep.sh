run() { echo "== $*"; env "$@" python tools/tlrun.py 2>&1 | grep ms/step; }
for i in 1 2; do
run A=1
run RAU_LIB=rau_vqa_amd/librau_r01.so
run RAU_ATT_SPLIT=1
run RAU_ENC_FUSED=1
run RAU_ATT_WAVES_BWD=8
done
