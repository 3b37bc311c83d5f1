run() { echo "== $*"; env "$@" python tools/tlrun.py 2>&1 | grep ms/step; }
run A=1
run RAU_SKINNY_WGS=256
run RAU_SKINNY_WGS=96
run RAU_HOP_GROUPS=3,2,2,1
run RAU_HOP_GROUPS=2,2,2,2
run RAU_HOP_GROUPS=4,2,2
run RAU_ATT_CHUNKS=8
run RAU_ATT_FUSED=1
run RAU_WGRAD_WGS=768
run RAU_GROUP_WGS=1024
run RAU_ENC_CHUNKS=3
run A=1
