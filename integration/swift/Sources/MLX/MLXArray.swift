// MLXArray.swift — module `MLX` of the shim: the part of mlx-swift's `MLX` module that code OUTSIDE Flux2Core uses (SURVEY §8b, row
// "MLXArray in the public API"), under the module name those files import.
//
// The reference hands `MLXArray`s across its public API (`Flux2StepHook`, `precomputedEmbeddings`, `initLatents`,
// `encodeImageToPackedSequence`, `packMaskForLatentBlending`) and the chains / CLI compute with them:
//   Flux2Chains/Flux2MaskedInpaintingChain.swift:18-19      @preconcurrency import MLX / MLXRandom
//   Flux2Chains/Flux2MaskedInpaintingChain.swift:380-403    MLXRandom.seed / normal, eval, MLXArray(Float), `1 - a`, `a * b`, `a + b`, .shape
//   Flux2CLI/InpaintCommand.swift:17, 253-271               import MLX; MLX.Memory.activeMemory / peakMemory / cacheMemory / clearCache()
// This module gives that code the same names over `flux2mi_array` handles and the library's allocation table, so it compiles and
// runs unchanged.  tests/test_swift_mlx_surface.py extracts every MLX name those files use (fixture tests/golden/mlx_surface_used.json)
// and requires a public declaration here for each.  `package` members are the seam to the `Flux2Core` target of this package.
import CFlux2MI

/// One process-wide context per GPU (the reference is one-generation-at-a-time too, Flux2Pipeline.swift:135).
public enum MIContext {
    public static var current: OpaquePointer = {
        var c: OpaquePointer?
        let s = flux2mi_ctx_create(0, nil, &c)
        precondition(s == 0, String(cString: flux2mi_last_error()))
        return c!
    }()
}

/// A non-zero `flux2mi_status` with the library's message; `Flux2Core.check(_:)` maps it onto `Flux2Error`.
package struct MIStatusError: Error {
    package let status: Int32
    package let message: String
}
@inline(__always) package func miCheck(_ status: Int32) throws {
    guard status != FLUX2MI_OK.rawValue else { return }
    throw MIStatusError(status: status, message: String(cString: flux2mi_last_error()))
}

public final class MLXArray: @unchecked Sendable {
    package let handle: OpaquePointer
    /// arrays this one was computed from: kept alive until the result is released (work may still be queued)
    private let inputs: [MLXArray]

    init(handle: OpaquePointer, inputs: [MLXArray] = []) { self.handle = handle; self.inputs = inputs }
    deinit { flux2mi_array_free(handle) }

    /// `MLXArray(Float)` — shape []
    public convenience init(_ value: Float) {
        var h: OpaquePointer?
        let s = flux2mi_array_from_scalar(MIContext.current, value, &h)
        precondition(s == 0, String(cString: flux2mi_last_error()))
        self.init(handle: h!)
    }
    public convenience init(_ value: Int) { self.init(Float(value)) }

    /// `MLXArray([Float], shape)`
    public convenience init(_ values: [Float], _ shape: [Int]) {
        var h: OpaquePointer?
        var sh = shape.map { Int64($0) }
        let s = values.withUnsafeBufferPointer { flux2mi_array_from_host(MIContext.current, $0.baseAddress, &sh, Int32(sh.count), &h) }
        precondition(s == 0, String(cString: flux2mi_last_error()))
        self.init(handle: h!)
    }

    /// Uninitialised owning array (outputs of encodeImageToPackedSequence / packMaskForLatentBlending, the initial noise's siblings).
    package static func uninitialized(_ shape: [Int]) throws -> MLXArray {
        var h: OpaquePointer?
        var sh = shape.map { Int64($0) }
        try miCheck(flux2mi_array_alloc(MIContext.current, &sh, Int32(sh.count), &h))
        return MLXArray(handle: h!)
    }

    /// Non-owning view over device memory (the hook's latent buffer, an encoder output).
    package static func borrowing(_ ptr: UnsafeMutablePointer<Float>, shape: [Int]) -> MLXArray {
        var h: OpaquePointer?
        var sh = shape.map { Int64($0) }
        let s = flux2mi_array_wrap(MIContext.current, ptr, &sh, Int32(sh.count), &h)
        precondition(s == 0, String(cString: flux2mi_last_error()))
        return MLXArray(handle: h!)
    }

    public var shape: [Int] { (0..<Int(flux2mi_array_ndim(handle))).map { Int(flux2mi_array_dim(handle, Int32($0))) } }
    package var data: UnsafeMutablePointer<Float> { flux2mi_array_data(handle) }
    package var count: Int { shape.reduce(1, *) }

    public func asArray() -> [Float] {
        var out = [Float](repeating: 0, count: count)
        let s = out.withUnsafeMutableBufferPointer { flux2mi_array_to_host(MIContext.current, handle, $0.baseAddress) }
        precondition(s == 0, String(cString: flux2mi_last_error()))
        return out
    }

    typealias BinOp = @convention(c) (OpaquePointer?, OpaquePointer?, OpaquePointer?, UnsafeMutablePointer<OpaquePointer?>?) -> Int32
    static func binary(_ op: BinOp, _ a: MLXArray, _ b: MLXArray) -> MLXArray {
        var h: OpaquePointer?
        let s = op(MIContext.current, a.handle, b.handle, &h)
        precondition(s == 0, String(cString: flux2mi_last_error()))   // MLX traps on non-broadcastable shapes as well
        return MLXArray(handle: h!, inputs: [a, b])
    }
}

// The operators the chains use (`(1 - sigmaNext) * imageLatents + sigmaNext * blendNoise`, `(1 - mask) * a + mask * b`):
// array (op) array with broadcasting, and numeric literals on either side.
public func + (a: MLXArray, b: MLXArray) -> MLXArray { MLXArray.binary(flux2mi_array_add, a, b) }
public func - (a: MLXArray, b: MLXArray) -> MLXArray { MLXArray.binary(flux2mi_array_sub, a, b) }
public func * (a: MLXArray, b: MLXArray) -> MLXArray { MLXArray.binary(flux2mi_array_mul, a, b) }
public func + (a: Float, b: MLXArray) -> MLXArray { MLXArray(a) + b }
public func + (a: MLXArray, b: Float) -> MLXArray { a + MLXArray(b) }
public func - (a: Float, b: MLXArray) -> MLXArray { MLXArray(a) - b }
public func - (a: MLXArray, b: Float) -> MLXArray { a - MLXArray(b) }
public func * (a: Float, b: MLXArray) -> MLXArray { MLXArray(a) * b }
public func * (a: MLXArray, b: Float) -> MLXArray { a * MLXArray(b) }
public func + (a: Int, b: MLXArray) -> MLXArray { MLXArray(Float(a)) + b }
public func - (a: Int, b: MLXArray) -> MLXArray { MLXArray(Float(a)) - b }
public func * (a: Int, b: MLXArray) -> MLXArray { MLXArray(Float(a)) * b }

/// `eval(_:)`: every op above is already enqueued on the context's stream; evaluation = stream synchronisation.
public func eval(_ arrays: MLXArray...) { flux2mi_ctx_sync(MIContext.current) }

/// `MLX.Memory` (mlx-swift 0.31.6 `Memory`; callers: Flux2CLI/InpaintCommand.swift:253-255, 271 and CompareEncoders.swift:112, 142):
/// the numbers come from the library's own allocation table (flux2mi_ctx_memory_stats) — active = weights, arrays, buffers, KV caches;
/// cache = the grow-only workspaces kept between calls; peak = the largest active + cache seen.  `clearCache()` = flux2mi_ctx_trim:
/// callable from any thread like MLX's — it gives nothing back while a library call is using the GPU's workspaces (flux2mi.h).
public enum Memory {
    static func stats() -> (active: Int, peak: Int, cache: Int) {
        var a = 0, p = 0, c = 0
        let s = flux2mi_ctx_memory_stats(MIContext.current, &a, &p, &c)
        precondition(s == 0, String(cString: flux2mi_last_error()))
        return (a, p, c)
    }
    public static var activeMemory: Int { stats().active }
    public static var cacheMemory: Int { stats().cache }
    public static var peakMemory: Int {
        get { stats().peak }
        set { flux2mi_ctx_reset_peak(MIContext.current) }     // mlx-swift: assigning (0) resets the high-water mark
    }
    public static func clearCache() { flux2mi_ctx_trim(MIContext.current, nil) }
}

/// `MLXRandom.seed(_:)` + keyless `MLXRandom.normal(_:)` — MLX's Threefry stream: the global key is split on every
/// draw, (state, sub) = split(state), and the draw uses `sub` (flux2mi.h, "MLX-compatible random stream").
public enum MLXRandom {
    static var state: (UInt32, UInt32) = (0, 0)
    public static func seed(_ seed: UInt64) {
        var k: (UInt32, UInt32) = (0, 0)
        withUnsafeMutablePointer(to: &k) { $0.withMemoryRebound(to: UInt32.self, capacity: 2) { flux2mi_mlx_random_key(seed, $0) } }
        state = k
    }
    public static func normal(_ shape: [Int]) -> MLXArray {
        var first: (UInt32, UInt32) = (0, 0), second: (UInt32, UInt32) = (0, 0), cur = state
        withUnsafeMutablePointer(to: &cur) { c in c.withMemoryRebound(to: UInt32.self, capacity: 2) { cp in
            withUnsafeMutablePointer(to: &first) { f in f.withMemoryRebound(to: UInt32.self, capacity: 2) { fp in
                withUnsafeMutablePointer(to: &second) { s in s.withMemoryRebound(to: UInt32.self, capacity: 2) { sp in
                    flux2mi_mlx_random_split(cp, fp, sp) } } } } } }
        state = first
        var h: OpaquePointer?
        var sh = shape.map { Int64($0) }
        let st = withUnsafeMutablePointer(to: &second) { s in s.withMemoryRebound(to: UInt32.self, capacity: 2) { sp in
            flux2mi_array_random_normal(MIContext.current, sp, &sh, Int32(sh.count), &h) } }
        precondition(st == 0, String(cString: flux2mi_last_error()))
        return MLXArray(handle: h!)
    }
}
