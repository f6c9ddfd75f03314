// Flux2Profiler.swift — the phase timer the chains and the CLI call (`Flux2Profiler.shared.enable() / reset() / start(_:) / end(_:) /
// generateReport()`: Flux2Chains/Flux2MaskedInpaintingChain.swift, Flux2CLI/InpaintCommand.swift:268-282, Flux2CLI/Flux2CLI.swift:141, 536, 712).
// In the reference `Flux2Profiler` is a typealias of the external swift-mlx-profiler package's `MLXProfiler`
// (Flux2Core/Utils/Flux2Profiler.swift:9), which sits on mlx-swift's `MLX` module — the module this package replaces — so the shim
// carries the small part of its surface those callers use: wall-clock phases, begun and ended on the calling thread.  Device time of the
// two phases that matter comes back from the library itself (`flux2mi_generate_params.ms_denoise / ms_decode`, HIP events on the
// generating stream) and is recorded by `Flux2Pipeline.generateWithResult` under the reference's phase names
// ("5. Denoising Loop", "6. VAE Decoding": Pipeline/Flux2Pipeline.swift:1933-2098).
import Foundation

public final class Flux2Profiler: @unchecked Sendable {
    public static let shared = Flux2Profiler()
    private let lock = NSLock()
    private var enabled = false
    private var open: [String: Date] = [:]
    private var timings: [(name: String, duration: Double)] = []

    public var isEnabled: Bool { lock.lock(); defer { lock.unlock() }; return enabled }
    public func enable() { lock.lock(); enabled = true; lock.unlock() }
    public func disable() { lock.lock(); enabled = false; lock.unlock() }
    public func reset() { lock.lock(); open.removeAll(); timings.removeAll(); lock.unlock() }

    public func start(_ name: String) {
        lock.lock(); defer { lock.unlock() }
        guard enabled else { return }
        open[name] = Date()
    }
    public func end(_ name: String) {
        lock.lock(); defer { lock.unlock() }
        guard enabled, let t0 = open.removeValue(forKey: name) else { return }
        timings.append((name, Date().timeIntervalSince(t0)))
    }
    /// A phase measured elsewhere (device milliseconds from the library's HIP events).
    public func record(_ name: String, seconds: Double) {
        lock.lock(); defer { lock.unlock() }
        guard enabled else { return }
        timings.append((name, seconds))
    }
    public func generateReport() -> String {
        lock.lock(); defer { lock.unlock() }
        guard !timings.isEmpty else { return "" }
        let total = timings.reduce(0.0) { $0 + $1.duration }
        var lines = ["", "FLUX.2 PERFORMANCE REPORT (libflux2mi)", "PHASE TIMINGS:"]
        for t in timings {
            let pct = total > 0 ? t.duration / total * 100 : 0
            lines.append(String(format: "  %@  %8.1f ms  %5.1f %%", t.name.padding(toLength: 36, withPad: " ", startingAt: 0), t.duration * 1000, pct))
        }
        lines.append(String(format: "  TOTAL  %.1f ms", total * 1000))
        return lines.joined(separator: "\n")
    }
}
