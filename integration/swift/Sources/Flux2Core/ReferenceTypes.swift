// ReferenceTypes.swift — stand-ins for the MLX-free declarations of Flux2Core that `Flux2Pipeline`'s signatures name, so that this
// package is self-contained.  INSIDE THE REFERENCE'S TREE THIS FILE IS NOT ADDED: the originals stay where they are —
//   Flux2Error                         Flux2Core.swift:14-40
//   Flux2Model                         Configuration/Flux2Config.swift:9-26
//   Flux2QuantizationConfig,           Configuration/QuantizationConfig.swift:40-110
//   TransformerQuantization, MistralQuantization
//   MemoryOptimizationConfig           Configuration/MemoryOptimizationConfig.swift
//   MemoryConfig.CacheProfile          Configuration/MemoryConfig.swift
//   ModelRegistry.VAEVariant, .ModelComponent   Configuration/ModelRegistry.swift:321-340
//   Flux2DownloadProgressCallback      Loading/ModelDownloader.swift:7
//   SchedulerOverrides, LoRAConfig, LoRAInfo    LoRA/LoRAConfig.swift:7-121
// Only what the pipeline's public surface touches is declared here, with the reference's names, cases and member names.
import CFlux2MI
import Foundation
import MLX

/// Flux2Core.swift:9-11 (`Flux2CLI.swift:25, 851` print it)
public enum Flux2Core {
    public static let version = "2.4.0"
}

public enum Flux2Error: Error, LocalizedError {
    case modelNotLoaded(String)
    case invalidConfiguration(String)
    case insufficientMemory(required: Int, available: Int)
    case weightLoadingFailed(String)
    case imageProcessingFailed(String)
    case generationFailed(String)
    case generationCancelled
}

/// Maps every status of `enum flux2mi_status` onto the reference's error type (Flux2Core.swift:14-40).
@inline(__always) func check(_ status: Int32) throws {
    guard status != FLUX2MI_OK.rawValue else { return }
    let msg = String(cString: flux2mi_last_error())
    switch flux2mi_status(rawValue: UInt32(status)) {
    case FLUX2MI_ERR_INVALID_ARG:  throw Flux2Error.invalidConfiguration(msg)
    case FLUX2MI_ERR_HIP:          throw Flux2Error.generationFailed(msg)
    case FLUX2MI_ERR_NOT_LOADED:   throw Flux2Error.modelNotLoaded(msg)
    case FLUX2MI_ERR_SHAPE:        throw Flux2Error.invalidConfiguration(msg)
    case FLUX2MI_ERR_CANCELLED:    throw Flux2Error.generationCancelled
    case FLUX2MI_ERR_UNSUPPORTED:  throw Flux2Error.invalidConfiguration(msg)
    case FLUX2MI_ERR_OOM:          throw Flux2Error.insufficientMemory(required: 0, available: 0)
    case FLUX2MI_ERR_UNKNOWN_KEY:  throw Flux2Error.weightLoadingFailed(msg)
    default:                       throw Flux2Error.generationFailed(msg)
    }
}
/// The same for errors thrown inside module `MLX` of this package (array allocation).
func rethrowing<T>(_ body: () throws -> T) throws -> T {
    do { return try body() } catch let e as MIStatusError { try check(e.status); fatalError("unreachable") }
}

public enum Flux2Model: String, CaseIterable, Sendable {
    case dev = "dev"
    case klein4B = "klein-4b"
    case klein4BBase = "klein-4b-base"
    case klein9B = "klein-9b"
    case klein9BBase = "klein-9b-base"
    case klein9BKV = "klein-9b-kv"
}

public enum TransformerQuantization: String, Codable, Sendable { case bf16, qint8, int4 }
public enum MistralQuantization: String, Codable, Sendable { case bf16, mlx8bit, mlx6bit, mlx4bit }
public struct Flux2QuantizationConfig: Codable, Sendable {
    public var textEncoder: MistralQuantization
    public var transformer: TransformerQuantization
    public init(textEncoder: MistralQuantization, transformer: TransformerQuantization) {
        self.textEncoder = textEncoder; self.transformer = transformer
    }
    public var estimatedTotalMemoryGB: Int { transformer == .bf16 ? 72 : 40 }
    public static let balanced = Flux2QuantizationConfig(textEncoder: .mlx8bit, transformer: .qint8)
    public static let highQuality = Flux2QuantizationConfig(textEncoder: .bf16, transformer: .bf16)
}
public struct MemoryOptimizationConfig: Sendable {
    public init() {}
    public static func recommended(forRAMGB: Int) -> MemoryOptimizationConfig { MemoryOptimizationConfig() }
}
public enum MemoryConfig {
    public enum CacheProfile: String, Sendable { case auto, conservative, performance }   // Configuration/MemoryConfig.swift
}
public enum ModelRegistry {
    public enum VAEVariant: String, CaseIterable, Sendable {
        case standard = "standard"
        case smallDecoder = "small-decoder"
    }
    public enum ModelComponent: Sendable {
        case transformer(Flux2Model)
        case textEncoder(Flux2Model)
        case vae(VAEVariant)
    }
}
public typealias Flux2DownloadProgressCallback = @Sendable (Double, String) -> Void

public struct SchedulerOverrides: Sendable, Codable, Equatable {
    public var customSigmas: [Float]?
    public var numSteps: Int?
    public var guidance: Float?
    public init(customSigmas: [Float]? = nil, numSteps: Int? = nil, guidance: Float? = nil) {
        self.customSigmas = customSigmas; self.numSteps = numSteps; self.guidance = guidance
    }
    public var hasOverrides: Bool { customSigmas != nil || numSteps != nil || guidance != nil }
}
public struct LoRAConfig: Sendable, Codable {
    public var filePath: String
    public var scale: Float?
    public var activationKeyword: String?
    public var schedulerOverrides: SchedulerOverrides?
    public var name: String { URL(fileURLWithPath: filePath).deletingPathExtension().lastPathComponent }
    public var effectiveScale: Float { scale ?? 1.0 }
    public init(filePath: String, scale: Float? = 1.0, activationKeyword: String? = nil, schedulerOverrides: SchedulerOverrides? = nil) {
        self.filePath = filePath; self.scale = scale; self.activationKeyword = activationKeyword; self.schedulerOverrides = schedulerOverrides
    }
    /// JSON sidecar of a LoRA (`Flux2CLI.swift:169, 500`): the Codable form of this struct
    public static func load(from path: String) throws -> LoRAConfig {
        try JSONDecoder().decode(LoRAConfig.self, from: Data(contentsOf: URL(fileURLWithPath: path)))
    }
    public func save(to path: String) throws {
        let enc = JSONEncoder()
        enc.outputFormatting = [.prettyPrinted, .sortedKeys]
        try enc.encode(self).write(to: URL(fileURLWithPath: path))
    }
}
public struct LoRAInfo: Sendable {
    public let numLayers: Int
    public let rank: Int
    public let numParameters: Int
    public enum TargetModel: String, Sendable { case klein4B = "klein-4b", klein9B = "klein-9b", dev = "dev", unknown = "unknown" }
    public let targetModel: TargetModel
}

/// Where the checkpoints of a component are on disk.  In the reference's tree: `Flux2ModelDownloader.findModelPath(for:)`
/// (Loading/ModelDownloader.swift) + the `*.safetensors` files under it; stand-alone: `$FLUX2MI_MODEL_DIR/<component>/`.
public enum MIModelLocator {
    public static func safetensors(for component: ModelRegistry.ModelComponent) -> [String] {
        let root = ProcessInfo.processInfo.environment["FLUX2MI_MODEL_DIR"] ?? "."
        let sub: String
        switch component {
        case .transformer(let m): sub = "transformer-\(m.rawValue)"
        case .textEncoder(let m): sub = "text-encoder-\(m.rawValue)"
        case .vae(let v): sub = "vae-\(v.rawValue)"
        }
        let dir = (root as NSString).appendingPathComponent(sub)
        let names = (try? FileManager.default.contentsOfDirectory(atPath: dir)) ?? []
        return names.filter { $0.hasSuffix(".safetensors") }.sorted().map { (dir as NSString).appendingPathComponent($0) }
    }
}

/// Prompt -> token ids of the text encoder's chat template.  In the reference's tree: FluxTextEncoders' tokenizer
/// (`KleinEmbeddingExtractor.swift:50-162` Klein: Qwen3 template, right-padded to 512; `EmbeddingExtractor.swift:202-278` Dev: Mistral
/// system + user message, left-padded) — host string work that stays Swift.  Returns exactly `maxLength` ids plus the range of real tokens.
public protocol Flux2PromptTokenizer: Sendable {
    func tokenize(prompt: String, maxLength: Int) throws -> (ids: [Int32], padLeft: Int, validLength: Int)
}
