"""gnnrag_amd - MI355X-native ReaRev reasoning hot path for GNN-RAG (gfx950 HIP)."""
__version__ = "0.1.0"
