"""Measurement harness behind bench.py (not product code)."""
