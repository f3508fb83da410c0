"""The secondary legs of bench.py (one JSON line is still produced by `python bench.py`)."""
