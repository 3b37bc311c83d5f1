run() { echo "== $*"; env "$@" python tools/tlrun.py 2>&1 | grep ms/step; }
for i in 1 2; do
run A=1
run RAU_BWD_GROUPS=1,1,1,1,1,1,1,1
run RAU_BWD_GROUPS=2,2,1,1,1,1
run RAU_BWD_GROUPS=2,2,2,2
run RAU_BWD_GROUPS=4,2,1,1
done
