set -e
for e in 2 3 4 6; do
  TVZ_CXXFLAGS="-DTVZ_MATCH_STEP=$e" python -m tvidz_amd.build --force > /dev/null 2>&1
  echo "STEP=$e"; python profiles/tune_match.py 100000 1024 2>&1 | grep '"min_match": 2' | head -1
  python profiles/tune_match.py 5000 1024 2>&1 | grep '"min_match": 2' | head -1
done
