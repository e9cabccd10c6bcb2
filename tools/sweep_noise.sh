for v in "VP_C3_AGRID_LIGHT=2" "VP_C3_AGRID_LIGHT=4" "VP_C3_AGRID_LIGHT=8" "VP_C3_AGRID_LIGHT=16" "VP_C3_LGRID=8" "VP_C3_LGRID=32" "VP_C3_BGRID=32" "VP_C3_BGRID=128" "VP_C3_AGRID=1" "VP_C3_AGRID=4"; do
  for lo in 230 128; do
    echo -n "$v lo=$lo: "; env $v python tools/exp_noise.py $lo 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels_us']; print(d['ms_per_step'], 'link',k['ccl_local'],'bound',k['ccl_boundary'],'rank',k['ccl_rank'],'label',k['ccl_stats'],'rows',k['ccl_final'])"
  done
done
