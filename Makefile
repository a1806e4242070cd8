# Convenience targets; the driver uses __graft_entry__.py, pytest and bench.py directly.
PY ?= python

.PHONY: build test-cpu test-gpu test-asan test-ubsan bench golden fuzz clean

build:            ## HIP library (gfx950), C oracle, C++ host mirror, zen CLI, C++ test program
	$(PY) -c "import __graft_entry__ as g; g.build()"

test-cpu: build   ## oracle vs reference vectors / fixtures, ABI, host logic, 2-process gloo
	$(PY) -m pytest tests -q -m "not gpu"

test-gpu: build   ## bit-exact parity through the C-ABI (needs an MI355X)
	$(PY) -m pytest tests -q -m gpu

test-asan:        ## oracle + WAV reader under ASAN+UBSAN here; add the C++ host mirror on a GPU box (pytest -m gpu)
	$(PY) -m pytest tests/test_sanitizers.py -q -m "not gpu"

test-ubsan:       ## the restatement alone under UBSAN (same driver, prints the checksum)
	$(MAKE) -s -C oracle san_driver_ubsan && oracle/san_driver_ubsan

bench: build      ## one JSON line: hops/s, roofline, cpu_baseline
	$(PY) bench.py

golden:           ## regenerate tests/golden/*.npz
	$(PY) tests/golden/make_golden.py

fuzz: build       ## randomised differential test against the oracle (needs an MI355X)
	$(PY) tools/fuzz_parity.py --seconds 120

clean:
	rm -rf zen_amd/build zen_amd/*.so zen_amd/bin oracle/*.so oracle/*.o oracle/san_driver_asan oracle/san_driver_ubsan tests/cpp/test_libzen tools/bin
