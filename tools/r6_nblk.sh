cd $GRAFT_REPO_ROOT
for v in "" "RG_NBLK=109" "RG_NBLK=37" "RG_NBLK=28"; do
  echo "== $v"
  env $v timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu --no-extra --no-disk 2>&1 | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], 'chol ms', d['kernels']['chol_f64']['ms'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])"
done
