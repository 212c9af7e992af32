"""Test helper: the Matrix-Market / tiling plumbing over the PRODUCT library's C ingest lives in primme_amd/ingest.py
(bench.py uses it too); the GPU cases of BASELINE configs[2] are fed through this path."""
from primme_amd.ingest import mm_read, tile_block_diagonal  # noqa: F401
