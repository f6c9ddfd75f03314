// swift-tools-version:5.9
// The two targets a maintainer of VincentGourbin/flux-2-swift-mlx adds to run the hot path on libflux2mi.so.
// NOT compiled in this repository's build image (no Swift toolchain there: INTEGRATION.md).
import PackageDescription

let package = Package(
    name: "Flux2MIShim",
    products: [.library(name: "Flux2MIShim", targets: ["Flux2MIShim"])],
    targets: [
        .systemLibrary(name: "CFlux2MI", path: "Sources/CFlux2MI"),
        .target(name: "Flux2MIShim", dependencies: ["CFlux2MI"], path: "Sources/Flux2MIShim"),
    ]
)
