# Convenience targets; __graft_entry__.build() does the same three builds.
#   make            libgenodsp_hip.so (hipcc, gfx950), the genodsp_hip driver (gcc), the test oracle (gcc)
#   make ref        also builds the unmodified reference from /root/reference into oracle/_ref (tests only)
#   make test       CPU tests (oracle vs golden vectors, ABI, two-rank gloo); GPU tests need -m gpu on an MI355X
all:
	$(MAKE) -C genodsp_amd/csrc
	$(MAKE) -C genodsp_amd/host
	$(MAKE) -C oracle

ref:
	$(MAKE) -C oracle ref

test: all
	python -m pytest tests -x -q -m "not gpu"

clean:
	$(MAKE) -C genodsp_amd/csrc clean
	$(MAKE) -C genodsp_amd/host clean
	$(MAKE) -C oracle clean

.PHONY: all ref test clean
