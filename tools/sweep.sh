run() { echo "== $*"; env "$@" python tools/tlrun.py 2>&1 | grep ms/step; }
for i in 1 2 3; do
run A=1
run RAU_LIB=rau_vqa_amd/librau_l6.so
done
