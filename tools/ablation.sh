B="timeout -k 10 240 python bench.py --steps 4 --warmup 1 --no-cpu-baseline"
run() { name=$1; shift; env "$@" $B > gpurun_out/abl_$name.json 2>/dev/null && tail -1 gpurun_out/abl_$name.json | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', d['value'], d['ms_per_step'])"; }
run base X=1 && run no_qstats DCAMD_NO_QSTATS=1 && run no_up4 DCAMD_NO_UP4=1 && run no_gn_wave DCAMD_GN_NO_WAVE=1 && run no_skip_split DCAMD_NO_SKIP_SPLIT=1 && run no_short_fold DCAMD_NO_SHORT_FOLD=1 && run no_ln_fold DCAMD_NO_LN_FOLD=1 && run no_thin DCAMD_NO_THIN=1 && run no_xreg DCAMD_NO_XREG=1 && run no_wide DCAMD_PIPE_NO_WIDE=1 && run no_halo DCAMD_NO_HALO=1 && run base2 X=1
timeout -k 10 240 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-share-trunk > gpurun_out/abl_no_trunk.json 2>/dev/null && tail -1 gpurun_out/abl_no_trunk.json | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('no_share_trunk', d['value'], d['ms_per_step'])"
run no_gn_ws DCAMD_NO_GN_WS=1 && run no_gn_out_fusion DCAMD_NO_GN_OUT_FUSION=1
run no_pn DCAMD_NO_PN=1 && run no_tblock DCAMD_NO_TBLOCK=1 && run no_po_fold DCAMD_NO_PO_FOLD=1 && run base3 X=1
