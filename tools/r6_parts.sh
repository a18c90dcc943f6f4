cd $GRAFT_REPO_ROOT
for p in 1 2 3 4; do
  echo "== RG_CHOL_PARTS=$p"
  RG_CHOL_PARTS=$p timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu --no-extra --no-disk 2>&1 | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], 'chol ms', d['kernels']['chol_f64']['ms'], d['roofline']['frac'], d['loco_checksum'])"
done
