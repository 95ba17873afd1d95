set -e
for e in 0 1 2 3; do
  TVZ_CXXFLAGS="-DTVZ_EXP=$e" python -m tvidz_amd.build --force > /dev/null 2>&1
  echo "EXP=$e"; python profiles/tune_match.py 100000 1024 2>&1 | grep '"min_match": 2' | head -1
done
