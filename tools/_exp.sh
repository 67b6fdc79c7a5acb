cd $GRAFT_REPO_ROOT
for hm in -1 1024 4096; do for w in 0 4 8; do
  echo "== heavy_min $hm waves $w"
  JR_FWD_HEAVY_MIN=$hm JR_FWD_HEAVY_WAVES=$w timeout 120 python tools/time_configs.py 2>&1 | grep -E "C4-like|C1 |C2 |280 faces|3 300"
done; done
