"""Formulation plug-ins (host side): offline ingredient computation per (formulation, method)."""
