from .vilt_module import ViLTransformerSS  # noqa: F401  (same export as the reference's vilt/modules/__init__.py:1)
