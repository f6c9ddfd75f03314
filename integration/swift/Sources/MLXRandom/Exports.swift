// Module `MLXRandom` of the shim: the reference imports it next to `MLX` (`@preconcurrency import MLXRandom`,
// Flux2Chains/Flux2MaskedInpaintingChain.swift:19) and calls `MLXRandom.seed(_:)` / `MLXRandom.normal(_:)` (:383-385).  As in mlx-swift
// 0.31.6 the functions live in the `MLXRandom` namespace of module `MLX`; this module re-exports it so that the import line compiles
// as written.
@_exported import MLX
