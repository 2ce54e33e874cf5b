# Convenience targets (the driver uses __graft_entry__.py / pytest / bench.py directly).
.PHONY: build test gpu-test bench clean
build:
	python -c "import __graft_entry__ as g; g.build()"
test: build
	python -m pytest tests -q -m "not gpu"
gpu-test: build          # needs an MI355X
	python -m pytest tests -q -m gpu
bench: build             # needs an MI355X
	python bench.py
clean:
	$(MAKE) -C pbs_plus_amd/csrc clean
	$(MAKE) -C oracle clean
