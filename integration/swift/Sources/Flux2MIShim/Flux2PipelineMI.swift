// Flux2PipelineMI.swift — `Flux2Pipeline.generateWithResult` (Pipeline/Flux2Pipeline.swift:1158-1176) over flux2mi_generate.
//
// Public types are the reference's own (Flux2Pipeline.swift:16-87, Flux2Core.swift:14-40), re-declared here only so that
// this package is self-contained; inside the reference's tree these declarations are deleted and the file joins Flux2Core.
import CFlux2MI
import CoreGraphics
import Foundation

public enum Flux2Error: Error {
    case modelNotLoaded(String), invalidConfiguration(String), insufficientMemory(required: Int, available: Int)
    case weightLoadingFailed(String), imageProcessingFailed(String), generationFailed(String), generationCancelled
}
public struct Flux2StepContext: Sendable {
    public let stepIdx: Int, totalSteps: Int
    public let sigma: Float, sigmaNext: Float
    public let height: Int, width: Int
    public let isI2I: Bool
}
public typealias Flux2StepHook = @Sendable (Flux2StepContext, MLXArray) -> MLXArray
public typealias Flux2ProgressCallback = @Sendable (Int, Int) -> Void
public typealias Flux2CheckpointCallback = @Sendable (Int, CGImage) -> Void

/// Swift closures cross the C boundary through an Unmanaged box, as Utils/MLXCheckpoint.swift:23-118 does.
final class CallbackBox {
    let onStep: Flux2StepHook?, onProgress: Flux2ProgressCallback?, onCheckpoint: Flux2CheckpointCallback?
    let seqLen: Int
    init(onStep: Flux2StepHook?, onProgress: Flux2ProgressCallback?, onCheckpoint: Flux2CheckpointCallback?, seqLen: Int) {
        self.onStep = onStep; self.onProgress = onProgress; self.onCheckpoint = onCheckpoint; self.seqLen = seqLen
    }
}

func makeCGImage(rgb: UnsafePointer<UInt8>, width: Int, height: Int) -> CGImage? {
    let data = Data(bytes: rgb, count: width * height * 3) as CFData
    guard let provider = CGDataProvider(data: data) else { return nil }
    return CGImage(width: width, height: height, bitsPerComponent: 8, bitsPerPixel: 24, bytesPerRow: width * 3,
                   space: CGColorSpaceCreateDeviceRGB(), bitmapInfo: CGBitmapInfo(rawValue: CGImageAlphaInfo.none.rawValue),
                   provider: provider, decode: nil, shouldInterpolate: false, intent: .defaultIntent)
}

public final class Flux2PipelineMI {
    let dit: OpaquePointer, vae: OpaquePointer
    public init(dit: OpaquePointer, vae: OpaquePointer) { self.dit = dit; self.vae = vae }

    /// The hot section of generateWithResult: text embeddings in (`precomputedEmbeddings`), CGImage out.
    /// `references`: already-encoded reference tokens of the imageToImage mode (encodeReferenceImages, :2143-2260).
    /// `embeddingsBF16`: device pointer to the prompt embeddings, bf16 [textLength, jointDim] (the output of
    /// flux2mi_textenc_forward, or `precomputedEmbeddings` converted once).
    public func generateWithResult(embeddingsBF16: UnsafeRawPointer, textLength: Int,
                                   noise: MLXArray, height: Int, width: Int, steps: Int, guidance: Float,
                                   initLatents: MLXArray? = nil, strength: Float = 1.0,
                                   references: (latents: MLXArray, ids: UnsafePointer<Int32>)? = nil,
                                   customSigmas: [Float]? = nil, checkpointInterval: Int? = nil,
                                   onProgress: Flux2ProgressCallback? = nil, onCheckpoint: Flux2CheckpointCallback? = nil,
                                   onStep: Flux2StepHook? = nil) throws -> CGImage {
        var vh: Int32 = 0, vw: Int32 = 0
        flux2mi_validate_dimensions(Int32(height), Int32(width), &vh, &vw)            // LatentUtils.validateDimensions
        let seqLen = Int(vh / 16) * Int(vw / 16)
        if let il = initLatents, il.shape != [1, seqLen, 128] {                        // :1898-1901
            throw Flux2Error.invalidConfiguration("initLatents shape \(il.shape) does not match latents shape [1, \(seqLen), 128]")
        }
        var p = flux2mi_generate_params()
        p.height = Int32(height); p.width = Int32(width); p.steps = Int32(steps); p.guidance = guidance
        p.txt_emb = embeddingsBF16; p.s_txt = Int32(textLength)
        p.noise = UnsafePointer(noise.data)
        p.init_latents = initLatents.map { UnsafePointer($0.data) }
        p.strength = strength
        if let r = references {
            p.ref_latents = UnsafePointer(r.latents.data); p.ref_ids = r.ids; p.s_ref = Int32(r.latents.shape[1])
        }
        p.checkpoint_interval = Int32(checkpointInterval ?? 0)

        let box = Unmanaged.passRetained(CallbackBox(onStep: onStep, onProgress: onProgress, onCheckpoint: onCheckpoint, seqLen: seqLen))
        defer { box.release() }
        let user = box.toOpaque()
        p.on_step_user = user; p.on_progress_user = user; p.on_checkpoint_user = user
        if onStep != nil {
            p.on_step = { cptr, latents, _, user in
                let b = Unmanaged<CallbackBox>.fromOpaque(user!).takeUnretainedValue()
                guard let hook = b.onStep, let c = cptr?.pointee, let latents = latents else { return 0 }
                let sc = Flux2StepContext(stepIdx: Int(c.step_idx), totalSteps: Int(c.total_steps), sigma: c.sigma,
                                          sigmaNext: c.sigma_next, height: Int(c.height), width: Int(c.width), isI2I: c.is_i2i != 0)
                // the hook sees the library's latent buffer as an MLXArray [1, S, 128] and RETURNS an array (:1999-2000):
                // a new one is copied back into the buffer, the same one is a no-op
                let result = hook(sc, MLXArray.borrowing(latents, shape: [1, b.seqLen, 128]))
                guard result.shape == [1, b.seqLen, 128] else { return 1 }
                return flux2mi_array_copy_to(MIContext.current, result.handle, latents, Int64(b.seqLen * 128)) == 0 ? 0 : 1
            }
        }
        if onProgress != nil {
            p.on_progress = { s, t, user in Unmanaged<CallbackBox>.fromOpaque(user!).takeUnretainedValue().onProgress?(Int(s), Int(t)) }
        }
        if onCheckpoint != nil {
            p.on_checkpoint = { step, rgb, h, w, user in
                if let rgb = rgb, let img = makeCGImage(rgb: rgb, width: Int(w), height: Int(h)) {
                    Unmanaged<CallbackBox>.fromOpaque(user!).takeUnretainedValue().onCheckpoint?(Int(step), img)
                }
            }
        }
        var image = [UInt8](repeating: 0, count: Int(vh) * Int(vw) * 3)
        let run: () throws -> Void = {
            try image.withUnsafeMutableBufferPointer { buf in
                p.out_image = buf.baseAddress
                try check(flux2mi_generate(MIContext.current, self.dit, self.vae, &p))
            }
        }
        if var cs = customSigmas {                                                     // LoRA scheduler override, :1879-1884
            try cs.withUnsafeMutableBufferPointer { b in
                p.custom_sigmas = UnsafePointer(b.baseAddress); p.n_custom_sigmas = Int32(b.count)
                try run()
            }
        } else {
            try run()
        }
        guard let cg = image.withUnsafeBufferPointer({ makeCGImage(rgb: $0.baseAddress!, width: Int(vw), height: Int(vh)) }) else {
            throw Flux2Error.imageProcessingFailed("Failed to create CGImage")          // postprocessVAEOutput, :2450-2467
        }
        return cg
    }
}

// The RePaint chain then runs AS WRITTEN in the reference (Flux2MaskedInpaintingChain.swift:380-403):
//
//     if let seed = seed { MLXRandom.seed(seed) }
//     let blendNoise = MLXRandom.normal(imageLatents.shape)
//     eval(blendNoise)
//     let onStep: Flux2StepHook = { ctx, latents in
//         let sigmaNext = MLXArray(ctx.sigmaNext)
//         let originalNoised = (1 - sigmaNext) * imageLatentsCaptured + sigmaNext * blendNoise
//         return (1 - maskLatentsCaptured) * originalNoised + maskLatentsCaptured * latents
//     }
//
// Six element-wise launches per step instead of one; a maintainer who wants the fused form replaces the closure body by
//     flux2mi_repaint_blend(MIContext.current, latents.data, imageLatentsCaptured.data, blendNoise.data,
//                           maskLatentsCaptured.data, ctx.sigmaNext, Int32(latents.shape[1]), 128); return latents
// — bit-identical results (tests/test_gpu_array_facade.py).
