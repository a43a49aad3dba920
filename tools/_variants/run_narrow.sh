set -e
for o in -1 0 1; do
  echo "== record_test=$o"
  RT_OPTIONS=record_test=$o timeout -k 10 200 python tools/share_time.py 1 temple 3840 2160 2>&1 | tail -1 | cut -c1-220
  RT_OPTIONS=record_test=$o timeout -k 10 300 python tools/share_time.py 1 ulica 1920 1080 2>&1 | tail -1 | cut -c1-220
done
