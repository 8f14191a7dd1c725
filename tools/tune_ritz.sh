# Round 6: sweep of the Ritz-harvest policy constants on configs[3] GP (GSFM_RITZ_* were temporary environment overrides of
# gp.hip while tuning; the shipped constants are static: this script documents how profiles/r06_gp_ritz_tuning.txt was made).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06l
run() { echo "== $*"; env "$@" python tools/exp_gp_recycle_gpu.py --only on 10000 1000000 0 2>&1 | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print(d['lm'], d['pcg'], d['ms_incl_h2d'], d['ritz_harvested'], d['final_cost'])
"; }
run A=1
run GSFM_RITZ_CUT=0.4
run GSFM_RITZ_CUT=0.2
run GSFM_RITZ_MINIT=20
run GSFM_RITZ_MINIT=30
run GSFM_RITZ_RATIO=2
run GSFM_RITZ_RATIO=5
run GSFM_RITZ_AGE=4
run GSFM_RITZ_AGE=10
run GSFM_RITZ_CONV=0.1
run GSFM_RITZ_CONV=0.5
