"""Namespace package: `penguin.jl_amd` is the MI355X-native implementation of Penguin.jl's hot path."""
