# Convenience targets (the driver uses __graft_entry__.py / pytest / bench.py directly).
.PHONY: build test gpu-test bench clean golden-go
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

# Golden vectors from the REAL Go module (github.com/pbs-plus/pxar v0.34.0) — needs Go + module access, neither of which the
# build image has (tools/golden/README.md). The fixture tests pick tests/golden/chunks_go.json up automatically.
golden-go:
	cd tools/golden && go mod tidy && go run . > ../../tests/golden/chunks_go.json.tmp
	mv tests/golden/chunks_go.json.tmp tests/golden/chunks_go.json
	@echo "wrote tests/golden/chunks_go.json - now: python -m pytest tests/test_oracle_buzhash.py -k golden -q"
