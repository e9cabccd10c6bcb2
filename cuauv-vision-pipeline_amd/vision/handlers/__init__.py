"""Handlers: per-object hooks a module hands its detections to (reference: core/handlers.py, handlers/)."""
