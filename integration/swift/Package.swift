// swift-tools-version:5.9
// The targets a maintainer of VincentGourbin/flux-2-swift-mlx swaps in to run the hot path on libflux2mi.so, under the MODULE NAMES the
// callers import — `import Flux2Core` (18 files of Flux2CLI + Flux2Chains), `import MLX`, `@preconcurrency import MLXRandom`
// (Flux2Chains/Flux2MaskedInpaintingChain.swift:17-19, Flux2CLI/InpaintCommand.swift:17) — so that those files compile as written:
//   in the reference's Package.swift the `Flux2Chains` / `Flux2CLI` targets keep their dependency lists; the products "MLX" and
//   "MLXRandom" of the mlx-swift package and the in-tree target "Flux2Core" are replaced by the three below (INTEGRATION.md §1).
// NOT compiled in this repository's build image (no Swift toolchain there: INTEGRATION.md).
import PackageDescription

let package = Package(
    name: "Flux2MI",
    products: [
        .library(name: "Flux2Core", targets: ["Flux2Core"]),
        .library(name: "MLX", targets: ["MLX"]),
        .library(name: "MLXRandom", targets: ["MLXRandom"]),
    ],
    targets: [
        .systemLibrary(name: "CFlux2MI", path: "Sources/CFlux2MI"),
        // MLXArray, operators, eval, Memory, MLXRandom (the namespace) over flux2mi_array / flux2mi_ctx_memory_stats
        .target(name: "MLX", dependencies: ["CFlux2MI"], path: "Sources/MLX"),
        // `import MLXRandom` compiles: re-exports MLX
        .target(name: "MLXRandom", dependencies: ["MLX"], path: "Sources/MLXRandom"),
        // Flux2Pipeline with the reference's public surface + the MLX-free types its signatures name
        .target(name: "Flux2Core", dependencies: ["CFlux2MI", "MLX"], path: "Sources/Flux2Core"),
    ]
)
