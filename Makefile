# Convenience targets (the driver uses __graft_entry__.py / pytest / bench.py directly).
PY ?= python

build:            ## hipcc --offload-arch=gfx950 -> tvidz_amd/libtvz.so, gcc -> oracle/libtvz_oracle.so
	$(PY) -c "import __graft_entry__ as g; g.build()"

test-cpu:         ## oracle vs golden fixtures, host logic, ABI, gloo sharding
	$(PY) -m pytest tests -x -q -m "not gpu"

test-gpu:         ## bit-exact parity through the C ABI (needs an MI355X)
	$(PY) -m pytest tests -x -q -m gpu

smoke:
	$(PY) -c "import __graft_entry__ as g; g.build(); g.smoke()"

bench:
	$(PY) bench.py

golden:           ## regenerate tests/golden from the reference (build container only)
	$(PY) oracle/gen_golden.py

.PHONY: build test-cpu test-gpu smoke bench golden
