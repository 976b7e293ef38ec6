"""Small host-side utilities shared by the caching / sparse configuration loaders."""
