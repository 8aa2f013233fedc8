"""The legs of bench.py (one module per family); bench.py parses, runs the headline and prints the ONE line."""
