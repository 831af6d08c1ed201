# the randomized soak under the library's A/B switches (each for MIN minutes; seeds differ)
MIN=${MIN:-4}
run() { echo "== $*"; env "$@" python tools/soak.py --minutes $MIN --seed $RANDOM 2>&1 | tail -2; }
run ESVIO_FE_X=0
run ESVIO_FE_WIDE_RECORDS=1
run ESVIO_FE_SAE_SORT=1
run ESVIO_FE_STAGE_THREADS=0
run ESVIO_FE_STAGE_THREADS=4
