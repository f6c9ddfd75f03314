// Flux2Pipeline.swift — `Flux2Pipeline` with the reference's own public surface (Sources/Flux2Core/Pipeline/Flux2Pipeline.swift,
// Flux2Pipeline+ChainHelpers.swift, v2.4.0), every body forwarding to the C ABI of libflux2mi.so (include/flux2mi.h).
//
// Drop-in rule: every `public` method below has EXACTLY the reference's name, argument labels, argument order, types, defaults,
// `async` / `throws` and return type, so that Flux2CLI (Flux2CLI.swift:338, 670), the chains (Flux2MaskedInpaintingChain.swift:303,
// 362, 368, 414-431; Flux2OutpaintingChain.swift) and the App compile against it unchanged.  tests/test_swift_shim_signatures.py parses
// the signatures out of the reference's two files and out of this one and requires equality.
//
//   reference declaration (file:line)                                   forwards to
//   init(model:quantization:memoryOptimization:vaeVariant:…)  :299       flux2mi_ctx_create (MIContext), presets
//   loadModels(progressCallback:)                              :370       flux2mi_{dit,vae,textenc}_create / _load_safetensors / _finalize
//   loadLoRA / unloadLoRA / unloadAllLoRAs / hasLoRA /         :618-689   flux2mi_dit_load_lora_safetensors
//     activeSchedulerOverrides
//   exportPrequantizedTransformer(allowLoRABaked:force:)       :759       flux2mi_dit_export_prequantized
//   generateTextToImage / generateImageToImage (x2)            :855-989   generate(mode:…)
//   generate…WithResult (x3)                                   :995-1101  generateWithResult(mode:…)
//   generate(mode:…onStep:)                                    :1104      generateWithResult(mode:…).image
//   generateWithResult(mode:…initLatents:strength:…onStep:)    :1158      flux2mi_textenc_forward, flux2mi_mlx_random_*, flux2mi_generate
//   clearAll / hasRequiredModels / missingModels               :2480-2523 flux2mi_*_destroy, MIModelLocator
//   encodeImageToPackedSequence(_:targetHeight:targetWidth:…)  +ChainHelpers:75   flux2mi_image_resample_u8, _preprocess_image_for_vae,
//                                                                                  flux2mi_vae_encode, flux2mi_vae_latents_to_packed
//   resolveChainDimensions(width:height:maxPixels:)            +ChainHelpers:112  flux2mi_resolve_chain_dimensions
//   packMaskForLatentBlending(_:targetHeight:…convention:)     +ChainHelpers:148  flux2mi_mask_inpaint_weights
//
// NOT compiled in this repository's build image (no Swift toolchain there: INTEGRATION.md).  What the reference does around the hot
// path and this file does not: prompt upsampling / `interpretImagePaths` (VLM generation: out of scope, §2 row 12 — the prompt is used
// as given, `wasUpsampled` is false), model download (`Flux2ModelDownloader`; checkpoints are looked up by MIModelLocator), Apple
// memory policy (`memoryOptimization`, `clearCacheEveryNSteps`, `compileDenoisingStep` are accepted and have no effect: they never
// change results, SURVEY §8a detail 12).
import CFlux2MI
import CoreGraphics
import Foundation
import ImageIO
import MLX

// MARK: - Public types of the generation API (Flux2Pipeline.swift:16-87)

public enum Flux2GenerationMode: Sendable {
    case textToImage
    case imageToImage(images: [CGImage])
}
public typealias Flux2ProgressCallback = @Sendable (Int, Int) -> Void
public typealias Flux2CheckpointCallback = @Sendable (Int, CGImage) -> Void
public struct Flux2StepContext: Sendable {
    public let stepIdx: Int
    public let totalSteps: Int
    public let sigma: Float
    public let sigmaNext: Float
    public let height: Int
    public let width: Int
    public let isI2I: Bool
}
public typealias Flux2StepHook = @Sendable (Flux2StepContext, MLXArray) -> MLXArray
public struct Flux2GenerationResult: Sendable {
    public let image: CGImage
    public let usedPrompt: String
    public let wasUpsampled: Bool
    public let originalPrompt: String
    public init(image: CGImage, usedPrompt: String, wasUpsampled: Bool, originalPrompt: String) {
        self.image = image; self.usedPrompt = usedPrompt; self.wasUpsampled = wasUpsampled; self.originalPrompt = originalPrompt
    }
}
public enum Flux2MaskConvention: Sendable, Equatable {
    case grayscaleWhiteInpaint
    case alphaTransparentInpaint
}

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

/// CGImage -> interleaved 8-bit pixels at the image's own size (no resampling: that happens on the device).
func pixels(of image: CGImage, channels: Int) -> [UInt8]? {
    let w = image.width, h = image.height
    var buf = [UInt8](repeating: 0, count: w * h * channels)
    let ok: Bool = buf.withUnsafeMutableBytes { raw in
        let space = channels == 1 ? CGColorSpaceCreateDeviceGray() : CGColorSpaceCreateDeviceRGB()
        let info = channels == 1 ? CGImageAlphaInfo.none.rawValue : CGImageAlphaInfo.premultipliedLast.rawValue
        guard let c = CGContext(data: raw.baseAddress, width: w, height: h, bitsPerComponent: 8, bytesPerRow: w * channels, space: space,
                                bitmapInfo: info) else { return false }
        c.interpolationQuality = .none
        c.draw(image, in: CGRect(x: 0, y: 0, width: w, height: h))
        return true
    }
    return ok ? buf : nil
}

/// A device buffer owned by Swift (flux2mi_buffer_alloc / _free).
final class DeviceBuffer {
    let ptr: UnsafeMutableRawPointer
    let bytes: Int
    init(bytes: Int) throws {
        var p: UnsafeMutableRawPointer?
        try check(flux2mi_buffer_alloc(MIContext.current, bytes, &p))
        self.ptr = p!; self.bytes = bytes
    }
    convenience init<T>(uploading host: [T]) throws {
        try self.init(bytes: host.count * MemoryLayout<T>.stride)
        try host.withUnsafeBytes { try check(flux2mi_buffer_upload(MIContext.current, ptr, $0.baseAddress, bytes)) }
    }
    deinit { flux2mi_buffer_free(MIContext.current, ptr) }
}

// MARK: - Flux2Pipeline

public final class Flux2Pipeline: @unchecked Sendable {
    public let model: Flux2Model
    public let quantization: Flux2QuantizationConfig
    public var memoryOptimization: MemoryOptimizationConfig
    public let vaeVariant: ModelRegistry.VAEVariant
    let kleinEncoderPath: URL?

    public private(set) var transformerLoadedFromPrequantized = false
    public private(set) var transformerHasMergedLoRAs = false
    public private(set) var isLoaded: Bool = false
    /// Accepted for source compatibility; Apple unified-memory policy with no effect on results (Flux2Pipeline.swift:150-280).
    public var memoryProfile: MemoryConfig.CacheProfile = .auto
    public var clearCacheEveryNSteps: Int = 5
    public var compileDenoisingStep: Bool = false
    public var keepTextEncoderLoaded: Bool = false
    /// Prompt -> token ids; set by the host application (FluxTextEncoders' tokenizer in the reference's tree).
    public var promptTokenizer: Flux2PromptTokenizer?

    var dit: OpaquePointer?, vae: OpaquePointer?, textEncoder: OpaquePointer?
    var jointDim = 0
    var loraNames: [String] = []
    var loraSchedulerOverrides: SchedulerOverrides?

    public init(
        model: Flux2Model = .dev,
        quantization: Flux2QuantizationConfig = .balanced,
        memoryOptimization: MemoryOptimizationConfig? = nil,
        vaeVariant: ModelRegistry.VAEVariant = .smallDecoder,
        hfToken: String? = nil,
        kleinEncoderPath: URL? = nil
    ) {
        self.model = model
        self.quantization = quantization
        self.vaeVariant = vaeVariant
        self.memoryOptimization = memoryOptimization ?? MemoryOptimizationConfig.recommended(forRAMGB: 288)
        self.kleinEncoderPath = kleinEncoderPath
        _ = MIContext.current
    }

    deinit {
        if let t = textEncoder { flux2mi_textenc_destroy(t) }
        if let d = dit { flux2mi_dit_destroy(d) }
        if let v = vae { flux2mi_vae_destroy(v) }
    }

    /// The architecture preset of a model variant (base / KV variants share their parent's: ModelRegistry, Flux2Config.swift:258-329).
    var presetName: String {
        switch model {
        case .dev: return "dev"
        case .klein4B, .klein4BBase: return "klein-4b"
        case .klein9B, .klein9BBase, .klein9BKV: return "klein-9b"
        }
    }

    // MARK: loading (Flux2Pipeline.swift:370-612, 692-757)

    public func loadModels(progressCallback: Flux2DownloadProgressCallback? = nil) async throws {
        if dit == nil {
            var cfg = flux2mi_dit_config()
            try check(flux2mi_dit_preset(presetName, &cfg))
            // a quantised transformer configuration runs the e4m3 block GEMMs (weight_format 1: DESIGN.md §4 — a different quantiser
            // from the reference's qint8 / int4, same role); bf16 stays bf16
            cfg.weight_format = quantization.transformer == .bf16 ? 0 : 1
            jointDim = Int(cfg.joint_dim)
            var d: OpaquePointer?
            try check(flux2mi_dit_create(MIContext.current, &cfg, &d))
            let files = MIModelLocator.safetensors(for: .transformer(model))
            guard !files.isEmpty else { flux2mi_dit_destroy(d); throw Flux2Error.modelNotLoaded("Transformer weights not found for \(model.rawValue)") }
            for (i, f) in files.enumerated() {
                var n: Int32 = 0, k: Int32 = 0
                try check(flux2mi_dit_load_safetensors(d, f, &n, &k))
                progressCallback?(Double(i + 1) / Double(files.count), "transformer")
            }
            try check(flux2mi_dit_finalize(d))
            dit = d
        }
        if vae == nil {
            var vcfg = flux2mi_vae_config()
            try check(flux2mi_vae_preset(vaeVariant == .standard ? "standard" : "small", &vcfg))
            var v: OpaquePointer?
            try check(flux2mi_vae_create(MIContext.current, &vcfg, &v))
            let files = MIModelLocator.safetensors(for: .vae(vaeVariant))
            guard !files.isEmpty else { flux2mi_vae_destroy(v); throw Flux2Error.modelNotLoaded("VAE weights not found for variant: \(vaeVariant.rawValue)") }
            for f in files {
                var n: Int32 = 0, k: Int32 = 0
                try check(flux2mi_vae_load_safetensors(v, f, &n, &k))
            }
            try check(flux2mi_vae_finalize(v))
            vae = v
            progressCallback?(1.0, "vae")
        }
        isLoaded = true
    }

    /// Phase "1. Load Text Encoder" (:1246-1262): Qwen3-4B / Qwen3-8B for Klein, the Mistral text model for Dev.
    func loadTextEncoderIfNeeded() throws {
        guard textEncoder == nil else { return }
        var tcfg = flux2mi_textenc_config()
        let name: String
        switch model {
        case .dev: name = "dev"                                  // Mistral-Small-3.2 text model, layers 10 / 20 / 30
        case .klein4B, .klein4BBase: name = "klein-4b"            // Qwen3-4B, layers 9 / 18 / 27
        case .klein9B, .klein9BBase, .klein9BKV: name = "klein-9b" // Qwen3-8B
        }
        try check(flux2mi_textenc_preset(name, &tcfg))
        var t: OpaquePointer?
        try check(flux2mi_textenc_create(MIContext.current, &tcfg, &t))
        var files = MIModelLocator.safetensors(for: .textEncoder(model))
        if let dir = kleinEncoderPath?.path, model != .dev {
            let names = (try? FileManager.default.contentsOfDirectory(atPath: dir)) ?? []
            files = names.filter { $0.hasSuffix(".safetensors") }.sorted().map { (dir as NSString).appendingPathComponent($0) }
        }
        guard !files.isEmpty else { flux2mi_textenc_destroy(t); throw Flux2Error.modelNotLoaded("Text encoder weights not found for \(model.rawValue)") }
        for f in files {
            var n: Int32 = 0, k: Int32 = 0
            try check(flux2mi_textenc_load_safetensors(t, f, &n, &k))
        }
        try check(flux2mi_textenc_finalize(t))
        textEncoder = t
    }

    // MARK: LoRA (Flux2Pipeline.swift:618-689)

    @discardableResult
    public func loadLoRA(_ config: LoRAConfig) throws -> LoRAInfo {
        guard let d = dit else { throw Flux2Error.modelNotLoaded("Transformer") }
        var merged: Int32 = 0, unmatched: Int32 = 0
        try check(flux2mi_dit_load_lora_safetensors(d, config.filePath, config.effectiveScale, &merged, &unmatched))
        loraNames.append(config.name)
        transformerHasMergedLoRAs = true
        if let overrides = config.schedulerOverrides, overrides.hasOverrides { loraSchedulerOverrides = overrides }
        let target: LoRAInfo.TargetModel
        switch model {
        case .dev: target = .dev
        case .klein4B, .klein4BBase: target = .klein4B
        case .klein9B, .klein9BBase, .klein9BKV: target = .klein9B
        }
        return LoRAInfo(numLayers: Int(merged), rank: 0, numParameters: 0, targetModel: target)
    }

    public func unloadLoRA(name: String) {
        // as in the reference (:666-672) weights already merged stay merged; the name is dropped, overrides cleared with the last one
        loraNames.removeAll { $0 == name }
        if loraNames.isEmpty { loraSchedulerOverrides = nil }
    }

    public func unloadAllLoRAs() {
        loraNames.removeAll()
        loraSchedulerOverrides = nil
    }

    public var hasLoRA: Bool {
        !loraNames.isEmpty
    }

    public var activeSchedulerOverrides: SchedulerOverrides? {
        loraSchedulerOverrides
    }

    @discardableResult
    public func exportPrequantizedTransformer(
        allowLoRABaked: Bool = false,
        force: Bool = false
    ) async throws -> URL {
        guard quantization.transformer != .bf16 else {
            throw Flux2Error.invalidConfiguration(
                "exportPrequantizedTransformer requires a quantized transformer configuration (current: bf16)")
        }
        try await loadModels()
        guard let d = dit else { throw Flux2Error.modelNotLoaded("Transformer") }
        if transformerHasMergedLoRAs && !allowLoRABaked {
            throw Flux2Error.invalidConfiguration("a LoRA is merged into the transformer; pass allowLoRABaked to export it baked in")
        }
        let root = ProcessInfo.processInfo.environment["FLUX2MI_MODEL_DIR"] ?? "."
        let url = URL(fileURLWithPath: root).appendingPathComponent("transformer-\(model.rawValue)/mlx-prequantized/\(quantization.transformer.rawValue)/transformer.safetensors")
        if FileManager.default.fileExists(atPath: url.path) && !force { return url }
        try FileManager.default.createDirectory(at: url.deletingLastPathComponent(), withIntermediateDirectories: true)
        try check(flux2mi_dit_export_prequantized(d, url.path, MIModelLocator.safetensors(for: .transformer(model)).joined(separator: ";")))
        return url
    }

    // MARK: - Generation API (Flux2Pipeline.swift:855-1176)

    public func generateTextToImage(
        prompt: String,
        interpretImagePaths: [String]? = nil,
        height: Int = 1024,
        width: Int = 1024,
        steps: Int = 50,
        guidance: Float = 4.0,
        seed: UInt64? = nil,
        upsamplePrompt: Bool = false,
        precomputedEmbeddings: MLXArray? = nil,
        checkpointInterval: Int? = nil,
        onProgress: Flux2ProgressCallback? = nil,
        onCheckpoint: Flux2CheckpointCallback? = nil
    ) async throws -> CGImage {
        try await generate(
            mode: .textToImage, prompt: prompt, interpretImagePaths: interpretImagePaths, height: height, width: width, steps: steps,
            guidance: guidance, seed: seed, upsamplePrompt: upsamplePrompt, precomputedEmbeddings: precomputedEmbeddings,
            checkpointInterval: checkpointInterval, onProgress: onProgress, onCheckpoint: onCheckpoint)
    }

    public func generateImageToImage(
        prompt: String,
        images: [CGImage],
        interpretImagePaths: [String]? = nil,
        height: Int? = nil,
        width: Int? = nil,
        steps: Int = 50,
        guidance: Float = 4.0,
        seed: UInt64? = nil,
        upsamplePrompt: Bool = false,
        checkpointInterval: Int? = nil,
        maxReferencePixels: Int = 1024 * 1024,
        onProgress: Flux2ProgressCallback? = nil,
        onCheckpoint: Flux2CheckpointCallback? = nil
    ) async throws -> CGImage {
        guard !images.isEmpty && images.count <= 3 else {
            throw Flux2Error.invalidConfiguration("Provide 1-3 reference images")
        }
        return try await generate(
            mode: .imageToImage(images: images), prompt: prompt, interpretImagePaths: interpretImagePaths,
            height: height ?? images[0].height, width: width ?? images[0].width, steps: steps, guidance: guidance, seed: seed,
            upsamplePrompt: upsamplePrompt, checkpointInterval: checkpointInterval, maxReferencePixels: maxReferencePixels,
            onProgress: onProgress, onCheckpoint: onCheckpoint)
    }

    public func generateImageToImage(
        prompt: String,
        imageData: [Data],
        interpretImagePaths: [String]? = nil,
        height: Int? = nil,
        width: Int? = nil,
        steps: Int = 50,
        guidance: Float = 4.0,
        seed: UInt64? = nil,
        upsamplePrompt: Bool = false,
        checkpointInterval: Int? = nil,
        maxReferencePixels: Int = 1024 * 1024,
        onProgress: Flux2ProgressCallback? = nil,
        onCheckpoint: Flux2CheckpointCallback? = nil
    ) async throws -> CGImage {
        let images = try Self.decode(imageData)
        return try await generateImageToImage(
            prompt: prompt, images: images, interpretImagePaths: interpretImagePaths, height: height, width: width, steps: steps,
            guidance: guidance, seed: seed, upsamplePrompt: upsamplePrompt, checkpointInterval: checkpointInterval,
            maxReferencePixels: maxReferencePixels, onProgress: onProgress, onCheckpoint: onCheckpoint)
    }

    public func generateTextToImageWithResult(
        prompt: String,
        interpretImagePaths: [String]? = nil,
        height: Int = 1024,
        width: Int = 1024,
        steps: Int = 50,
        guidance: Float = 4.0,
        seed: UInt64? = nil,
        upsamplePrompt: Bool = false,
        checkpointInterval: Int? = nil,
        onProgress: Flux2ProgressCallback? = nil,
        onCheckpoint: Flux2CheckpointCallback? = nil
    ) async throws -> Flux2GenerationResult {
        try await generateWithResult(
            mode: .textToImage, prompt: prompt, interpretImagePaths: interpretImagePaths, height: height, width: width, steps: steps,
            guidance: guidance, seed: seed, upsamplePrompt: upsamplePrompt, checkpointInterval: checkpointInterval,
            onProgress: onProgress, onCheckpoint: onCheckpoint)
    }

    public func generateImageToImageWithResult(
        prompt: String,
        images: [CGImage],
        interpretImagePaths: [String]? = nil,
        height: Int? = nil,
        width: Int? = nil,
        steps: Int = 50,
        guidance: Float = 4.0,
        seed: UInt64? = nil,
        upsamplePrompt: Bool = false,
        checkpointInterval: Int? = nil,
        maxReferencePixels: Int = 1024 * 1024,
        onProgress: Flux2ProgressCallback? = nil,
        onCheckpoint: Flux2CheckpointCallback? = nil
    ) async throws -> Flux2GenerationResult {
        guard !images.isEmpty && images.count <= 3 else {
            throw Flux2Error.invalidConfiguration("Provide 1-3 reference images")
        }
        return try await generateWithResult(
            mode: .imageToImage(images: images), prompt: prompt, interpretImagePaths: interpretImagePaths,
            height: height ?? images[0].height, width: width ?? images[0].width, steps: steps, guidance: guidance, seed: seed,
            upsamplePrompt: upsamplePrompt, checkpointInterval: checkpointInterval, maxReferencePixels: maxReferencePixels,
            onProgress: onProgress, onCheckpoint: onCheckpoint)
    }

    public func generateImageToImageWithResult(
        prompt: String,
        imageData: [Data],
        interpretImagePaths: [String]? = nil,
        height: Int? = nil,
        width: Int? = nil,
        steps: Int = 50,
        guidance: Float = 4.0,
        seed: UInt64? = nil,
        upsamplePrompt: Bool = false,
        checkpointInterval: Int? = nil,
        maxReferencePixels: Int = 1024 * 1024,
        onProgress: Flux2ProgressCallback? = nil,
        onCheckpoint: Flux2CheckpointCallback? = nil
    ) async throws -> Flux2GenerationResult {
        let images = try Self.decode(imageData)
        return try await generateImageToImageWithResult(
            prompt: prompt, images: images, interpretImagePaths: interpretImagePaths, height: height, width: width, steps: steps,
            guidance: guidance, seed: seed, upsamplePrompt: upsamplePrompt, checkpointInterval: checkpointInterval,
            maxReferencePixels: maxReferencePixels, onProgress: onProgress, onCheckpoint: onCheckpoint)
    }

    public func generate(
        mode: Flux2GenerationMode,
        prompt: String,
        interpretImagePaths: [String]? = nil,
        height: Int,
        width: Int,
        steps: Int,
        guidance: Float,
        seed: UInt64?,
        upsamplePrompt: Bool,
        precomputedEmbeddings: MLXArray? = nil,
        checkpointInterval: Int?,
        maxReferencePixels: Int = 1024 * 1024,
        onProgress: Flux2ProgressCallback?,
        onCheckpoint: Flux2CheckpointCallback?,
        onStep: Flux2StepHook? = nil
    ) async throws -> CGImage {
        let result = try await generateWithResult(
            mode: mode, prompt: prompt, interpretImagePaths: interpretImagePaths, height: height, width: width, steps: steps,
            guidance: guidance, seed: seed, upsamplePrompt: upsamplePrompt, precomputedEmbeddings: precomputedEmbeddings,
            checkpointInterval: checkpointInterval, maxReferencePixels: maxReferencePixels, onProgress: onProgress,
            onCheckpoint: onCheckpoint, onStep: onStep)
        return result.image
    }

    public func generateWithResult(
        mode: Flux2GenerationMode,
        prompt: String,
        interpretImagePaths: [String]? = nil,
        height: Int,
        width: Int,
        steps: Int,
        guidance: Float,
        seed: UInt64?,
        upsamplePrompt: Bool,
        precomputedEmbeddings: MLXArray? = nil,
        checkpointInterval: Int?,
        maxReferencePixels: Int = 1024 * 1024,
        initLatents: MLXArray? = nil,
        strength: Float = 1.0,
        onProgress: Flux2ProgressCallback?,
        onCheckpoint: Flux2CheckpointCallback?,
        onStep: Flux2StepHook? = nil
    ) async throws -> Flux2GenerationResult {
        // LatentUtils.validateDimensions (:1191)
        var vh: Int32 = 0, vw: Int32 = 0
        flux2mi_validate_dimensions(Int32(height), Int32(width), &vh, &vw)
        let seqLen = Int(vh / 16) * Int(vw / 16)
        if let seed = seed { MLXRandom.seed(seed) }                                      // :1203-1205
        let effectiveStrength = max(0.01, min(1.0, strength))                             // :1207-1213
        if effectiveStrength < 1.0 && initLatents == nil {
            throw Flux2Error.invalidConfiguration(
                "strength < 1.0 requires initLatents (packed-sequence latents of the init image, see encodeImageToPackedSequence)")
        }
        if let il = initLatents, il.shape != [1, seqLen, 128] {                           // :1898-1901
            throw Flux2Error.invalidConfiguration("initLatents shape \(il.shape) does not match latents shape [1, \(seqLen), 128]")
        }
        try await loadModels()
        guard let dit = dit else { throw Flux2Error.modelNotLoaded("Transformer") }
        guard let vae = vae else { throw Flux2Error.modelNotLoaded("VAE") }

        // PHASE 1-2, text: `precomputedEmbeddings` [1, S_txt, J] as given, else the text encoder on the prompt (:1228-1418)
        let textLength: Int
        let embeddings: DeviceBuffer
        if let pre = precomputedEmbeddings {
            let sh = pre.shape
            guard sh.count == 3, sh[0] == 1, sh[2] == jointDim else {
                throw Flux2Error.invalidConfiguration("precomputedEmbeddings shape \(sh), expected [1, S, \(jointDim)]")
            }
            textLength = sh[1]
            embeddings = try DeviceBuffer(bytes: textLength * jointDim * 2)
            try check(flux2mi_array_to_bf16(MIContext.current, pre.handle, embeddings.ptr))
        } else {
            guard let tok = promptTokenizer else { throw Flux2Error.modelNotLoaded("TextEncoder tokenizer (set promptTokenizer, or pass precomputedEmbeddings)") }
            try loadTextEncoderIfNeeded()
            textLength = 512                                                              // KleinEmbeddingExtractor.swift:78-87, MistralEncoder.swift:26
            let t = try tok.tokenize(prompt: prompt, maxLength: textLength)
            let ids = try DeviceBuffer(uploading: t.ids)
            embeddings = try DeviceBuffer(bytes: textLength * jointDim * 2)
            try check(flux2mi_textenc_forward(textEncoder, ids.ptr.assumingMemoryBound(to: Int32.self), Int32(textLength), Int32(t.padLeft),
                                              Int32(t.validLength), embeddings.ptr))
            if !keepTextEncoderLoaded { flux2mi_ctx_sync(MIContext.current); flux2mi_textenc_destroy(textEncoder); textEncoder = nil }   // :1418
        }

        // initial noise: LatentUtils.generatePatchifiedLatents re-seeds and draws [1,128,H/16,W/16] (LatentUtils.swift:32-41, :1448)
        if let seed = seed { MLXRandom.seed(seed) }
        let noise = MLXRandom.normal([1, 128, Int(vh / 16), Int(vw / 16)])

        // reference tokens of the imageToImage mode (encodeReferenceImages, :2143-2260)
        var references: (latents: MLXArray, ids: DeviceBuffer, count: Int)? = nil
        if case .imageToImage(let images) = mode {
            guard !images.isEmpty && images.count <= 3 else { throw Flux2Error.invalidConfiguration("Provide 1-3 reference images") }
            references = try await encodeReferenceImages(images, maxReferencePixels: maxReferencePixels)
        }

        var p = flux2mi_generate_params()
        p.height = Int32(height); p.width = Int32(width); p.steps = Int32(steps); p.guidance = guidance
        p.txt_emb = UnsafeRawPointer(embeddings.ptr); p.s_txt = Int32(textLength)
        p.noise = UnsafePointer(noise.data)
        p.init_latents = initLatents.map { UnsafePointer($0.data) }
        p.strength = strength
        if let r = references {
            p.ref_latents = UnsafePointer(r.latents.data)
            p.ref_ids = UnsafePointer(r.ids.ptr.assumingMemoryBound(to: Int32.self))
            p.s_ref = Int32(r.count)
            p.use_kv_cache = model == .klein9BKV ? 1 : 0                                  // :1555-1682
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
                try check(flux2mi_generate(MIContext.current, dit, vae, &p))
            }
        }
        if var cs = loraSchedulerOverrides?.customSigmas {                                // Turbo-style LoRAs bring their schedule, :1873-1884
            try cs.withUnsafeMutableBufferPointer { b in
                p.custom_sigmas = UnsafePointer(b.baseAddress); p.n_custom_sigmas = Int32(b.count)
                try run()
            }
        } else {
            try run()
        }
        // the two device phases under the reference's names (Flux2Pipeline.swift:1933-2052 "5. Denoising Loop", :2058-2098 "6. VAE Decoding"):
        // HIP-event milliseconds of the generating stream, reported by the library
        Flux2Profiler.shared.record("5. Denoising Loop", seconds: Double(p.ms_denoise) / 1000)
        Flux2Profiler.shared.record("6. VAE Decoding", seconds: Double(p.ms_decode) / 1000)
        guard let cg = image.withUnsafeBufferPointer({ makeCGImage(rgb: $0.baseAddress!, width: Int(vw), height: Int(vh)) }) else {
            throw Flux2Error.imageProcessingFailed("Failed to create CGImage")            // postprocessVAEOutput, :2450-2467
        }
        // prompt upsampling is VLM generation (out of scope): the prompt is used as given
        return Flux2GenerationResult(image: cg, usedPrompt: prompt, wasUpsampled: false, originalPrompt: prompt)
    }

    // MARK: - housekeeping (Flux2Pipeline.swift:2480-2523)

    public func clearAll() async {
        flux2mi_ctx_sync(MIContext.current)
        if let t = textEncoder { flux2mi_textenc_destroy(t); textEncoder = nil }
        if let d = dit { flux2mi_dit_destroy(d); dit = nil }
        if let v = vae { flux2mi_vae_destroy(v); vae = nil }
        transformerHasMergedLoRAs = false
        isLoaded = false
    }

    /// Resident weight bytes of the configuration in GB (the reference sums its per-component estimates, :2475-2477).
    public var estimatedMemoryGB: Int {
        quantization.estimatedTotalMemoryGB
    }

    public var hasRequiredModels: Bool {
        missingModels.isEmpty
    }

    public var missingModels: [ModelRegistry.ModelComponent] {
        var missing: [ModelRegistry.ModelComponent] = []
        for c in [ModelRegistry.ModelComponent.textEncoder(model), .transformer(model), .vae(vaeVariant)]
        where MIModelLocator.safetensors(for: c).isEmpty {
            missing.append(c)
        }
        return missing
    }

    // MARK: - helpers

    /// Pixel-exact decode of PNG / JPEG bytes (:2256-2262).
    public static func cgImage(from data: Data) -> CGImage? {
        guard let src = CGImageSourceCreateWithData(data as CFData, nil) else { return nil }
        return CGImageSourceCreateImageAtIndex(src, 0, nil)
    }

    static func decode(_ imageData: [Data]) throws -> [CGImage] {
        try imageData.enumerated().map { index, data in
            guard let cgImage = Self.cgImage(from: data) else {
                throw Flux2Error.invalidConfiguration("Failed to decode image data at index \(index)")
            }
            return cgImage
        }
    }

    /// `encodeReferenceImages` (:2143-2260): scale each image down to at most `maxReferencePixels`, floor to multiples of 32, VAE-encode
    /// (posterior mean), patchify, BN-normalise, pack; concatenate; position ids with T = 10, 20, ... per image (LatentUtils.swift:324-346).
    func encodeReferenceImages(_ images: [CGImage], maxReferencePixels: Int) async throws -> (latents: MLXArray, ids: DeviceBuffer, count: Int) {
        let maxArea = max(maxReferencePixels, 32 * 32)
        var lh: [Int32] = [], lw: [Int32] = [], parts: [MLXArray] = [], total = 0
        for img in images {
            var tw = img.width, th = img.height
            if tw * th > maxArea {
                let scale = (Double(maxArea) / Double(tw * th)).squareRoot()
                tw = Int(Double(tw) * scale); th = Int(Double(th) * scale)
            }
            tw = max((tw / 32) * 32, 32); th = max((th / 32) * 32, 32)
            parts.append(try await encodeImageToPackedSequence(img, targetHeight: th, targetWidth: tw))
            lh.append(Int32(th / 16)); lw.append(Int32(tw / 16))
            total += (th / 16) * (tw / 16)
        }
        var ids = [Int32](repeating: 0, count: total * 4)
        let n = flux2mi_reference_position_ids(&lh, &lw, Int32(images.count), 10, &ids, Int32(total))
        guard Int(n) == total else { throw Flux2Error.generationFailed("reference position ids") }
        // one contiguous [1, total, 128] array: the packed sequences back to back
        let all = try rethrowing { try MLXArray.uninitialized([1, total, 128]) }
        var off = 0
        for a in parts {
            try check(flux2mi_array_copy_to(MIContext.current, a.handle, all.data + off, Int64(a.count)))
            off += a.count
        }
        return (all, try DeviceBuffer(uploading: ids), total)
    }
}

// MARK: - Chain helpers (Flux2Pipeline+ChainHelpers.swift:32-236)

extension Flux2Pipeline {

    public func encodeImageToPackedSequence(
        _ image: CGImage,
        targetHeight: Int,
        targetWidth: Int,
        samplePosterior: Bool = false
    ) async throws -> MLXArray {
        try await loadModels()
        guard let vae = vae else {
            throw Flux2Error.modelNotLoaded("VAE")
        }
        guard flux2mi_vae_has_encoder(vae) != 0 else { throw Flux2Error.modelNotLoaded("VAE encoder") }
        guard !samplePosterior else { throw Flux2Error.invalidConfiguration("samplePosterior: the posterior mean is what the pipeline uses (Flux2Pipeline.swift:2201)") }
        guard let rgba = pixels(of: image, channels: 4) else { throw Flux2Error.imageProcessingFailed("CGContext") }
        let src = try DeviceBuffer(uploading: rgba)
        // the reference draws the image into a target-sized CGContext (preprocessImageForVAE, :2359-2418); here: resample on the device
        let resized = try DeviceBuffer(bytes: targetHeight * targetWidth * 4)
        try check(flux2mi_image_resample_u8(MIContext.current, src.ptr.assumingMemoryBound(to: UInt8.self), Int32(image.height), Int32(image.width), 4, nil,
                                            resized.ptr.assumingMemoryBound(to: UInt8.self), Int32(targetHeight), Int32(targetWidth)))
        let x = try rethrowing { try MLXArray.uninitialized([1, 3, targetHeight, targetWidth]) }
        try check(flux2mi_preprocess_image_for_vae(MIContext.current, resized.ptr.assumingMemoryBound(to: UInt8.self), Int32(targetHeight), Int32(targetWidth), 4, x.data))
        let z = try rethrowing { try MLXArray.uninitialized([1, 32, targetHeight / 8, targetWidth / 8]) }
        try check(flux2mi_vae_encode(vae, x.data, Int32(targetHeight), Int32(targetWidth), z.data))
        let packed = try rethrowing { try MLXArray.uninitialized([1, (targetHeight / 16) * (targetWidth / 16), 128]) }
        try check(flux2mi_vae_latents_to_packed(MIContext.current, z.data, flux2mi_vae_bn_mean(vae), flux2mi_vae_bn_var(vae), packed.data,
                                                Int32(targetHeight / 16), Int32(targetWidth / 16)))
        eval(packed)
        return packed
    }

    public static func resolveChainDimensions(
        width: Int,
        height: Int,
        maxPixels: Int = 1024 * 1024
    ) -> (height: Int, width: Int) {
        var h: Int32 = 0, w: Int32 = 0
        flux2mi_resolve_chain_dimensions(Int32(width), Int32(height), Int32(maxPixels), &h, &w)
        return (height: Int(h), width: Int(w))
    }

    public static func packMaskForLatentBlending(
        _ mask: CGImage,
        targetHeight: Int,
        targetWidth: Int,
        convention: Flux2MaskConvention = .grayscaleWhiteInpaint
    ) async -> MLXArray {
        let latentH = targetHeight / 16
        let latentW = targetWidth / 16
        let count = latentH * latentW
        let channels = convention == .grayscaleWhiteInpaint ? 1 : 4
        let conv = convention == .grayscaleWhiteInpaint ? FLUX2MI_MASK_GRAYSCALE_WHITE_INPAINT : FLUX2MI_MASK_ALPHA_TRANSPARENT_INPAINT
        // the reference falls back to an all-zero mask when the CGContext cannot be made (:178-180, :222-224)
        guard let px = pixels(of: mask, channels: channels), let src = try? DeviceBuffer(uploading: px),
              let out = try? MLXArray.uninitialized([1, count, 1]),
              flux2mi_mask_inpaint_weights(MIContext.current, src.ptr.assumingMemoryBound(to: UInt8.self), Int32(mask.height), Int32(mask.width),
                                           Int32(channels), Int32(conv.rawValue), nil, out.data, Int32(latentH), Int32(latentW)) == 0
        else {
            return MLXArray([Float](repeating: 0, count: count), [1, count, 1])
        }
        return out
    }
}

// The RePaint chain then runs AS WRITTEN in the reference (Flux2MaskedInpaintingChain.swift:380-431):
//
//     if let seed = seed { MLXRandom.seed(seed) }
//     let blendNoise = MLXRandom.normal(imageLatents.shape)
//     eval(blendNoise)
//     let onStep: Flux2StepHook = { ctx, latents in
//         let sigmaNext = MLXArray(ctx.sigmaNext)
//         let originalNoised = (1 - sigmaNext) * imageLatentsCaptured + sigmaNext * blendNoise
//         return (1 - maskLatentsCaptured) * originalNoised + maskLatentsCaptured * latents
//     }
//     let result = try await pipeline.generateWithResult(mode: mode, prompt: resolvedPrompt, interpretImagePaths: nil, height: targetH,
//         width: targetW, steps: steps, guidance: guidance, seed: seed, upsamplePrompt: resolvedUpsample, precomputedEmbeddings: nil,
//         checkpointInterval: nil, initLatents: strength < 1.0 ? imageLatents : nil, strength: strength, onProgress: onProgress,
//         onCheckpoint: nil, onStep: onStep)
//
// Six element-wise launches per step instead of one; a maintainer who wants the fused form replaces the closure body by
//     flux2mi_repaint_blend(MIContext.current, latents.data, imageLatentsCaptured.data, blendNoise.data,
//                           maskLatentsCaptured.data, ctx.sigmaNext, Int32(latents.shape[1]), 128); return latents
// — bit-identical results (tests/test_gpu_array_facade.py).
