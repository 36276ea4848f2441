"""bench.py split per workload (one module each); bench.py itself is the entry point, the argument parser and the launcher."""
