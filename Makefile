# Convenience targets (the driver uses __graft_entry__.build() / pytest / bench.py directly).
.PHONY: build test test-gpu bench tools clean
build:
	python -c "import __graft_entry__ as g; g.build()"
test: build
	python -m pytest tests -x -q -m "not gpu"
test-gpu: build
	python -m pytest tests -x -q -m gpu
bench: build
	python bench.py
tools:
	$(MAKE) -C tools
clean:
	$(MAKE) -C slam-loop-closing_amd/csrc clean
	$(MAKE) -C oracle clean
	$(MAKE) -C tools clean
