for cfg in "0 0 -" "1024 16 -" "1024 8 -" "512 16 dense_kd_nt=1" "1024 16 dense_kd_nt=1" "256 16 -" "512 8 -"; do
  set -- $cfg; extra=""; [ "$3" = "-" ] || extra="--tune $3"
  python bench.py --dense-only --dense-size 256 --dense-tile-quads $1 --dense-tile-planes $2 $extra 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']
print('256^3 T=$1 zc=$2 $3 : KD %.2f us (%.3f)  KU %.2f us (%.3f)' % (k['pcg_dir']['avg_us'], k['pcg_dir']['frac'], k['pcg_update']['avg_us'], k['pcg_update']['frac']))"
done
for cfg in "0 0 -" "512 32 dense_kd_nt=1" "1024 32 -" "512 64 -"; do
  set -- $cfg; extra=""; [ "$3" = "-" ] || extra="--tune $3"
  python bench.py --dense-only --dense-size 512 --dense-tile-quads $1 --dense-tile-planes $2 $extra 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']
print('512^3 T=$1 zc=$2 $3 : KD %.2f us (%.3f)  KU %.2f us (%.3f)' % (k['pcg_dir']['avg_us'], k['pcg_dir']['frac'], k['pcg_update']['avg_us'], k['pcg_update']['frac']))"
done
