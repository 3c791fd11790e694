"""The parts of bench.py (workloads, verification, CPU baselines, PMC passes, the multi-rank program) and the
stand-alone measurement scripts of the rounds."""
