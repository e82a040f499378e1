//! # QOA - Quite OK Audio Format, MI355X backend
//!
//! UNCOMPILED SOURCE (no Rust toolchain where this repository is built).  Same public items as
//! `rafaelcaricio/qoaudio` 1.0 (`src/lib.rs` of the reference; line numbers below refer to it), forwarded to
//! the C ABI in `include/qoaudio_amd.h`.  The reference forbids unsafe code; an FFI shim cannot.
//!
//! Round 5: the `extern "C"` block binds EVERY non-debug symbol of the header (tests/test_rust_shim_abi.py fails when one is
//! missing) and the batch entry points have safe wrappers (`EncodePlan`, `DecodePlan`, `HostPipeline`, `stream_scan`).
//!
//! Every public signature equals the reference's (`mode()` and `current_frame_header()` return references to values
//! cached in the handle, as `src/lib.rs:180-188` do).  tests/test_rust_shim_abi.py checks the `extern "C"` block below
//! against `include/qoaudio_amd.h` symbol by symbol (argument count, pointer depth, constness, integer width) -- the
//! only verification possible without `rustc`.
//!
//! Differences a caller can observe: `QoaDecoder::new(reader)` pulls the reader lazily but a
//! whole FRAME at a time (the reference reads it slice row by slice row), so `into_inner()` in the middle of a frame
//! returns the reader at that frame's end; there is no CPU fallback (no usable GPU => `IoError`).
use std::ffi::CStr;
use std::fmt::Display;
use std::fs::File;
use std::io::Cursor;
use std::os::raw::{c_char, c_double, c_int, c_void};
use std::path::Path;
use std::time::Duration;
use std::{fmt, io};

pub const QOA_SLICE_LEN: usize = 20; // :12
pub const QOA_LMS_LEN: usize = 4; // :13
pub const QOA_HEADER_SIZE: usize = 8; // :14
pub const QOA_MAGIC: u32 = u32::from_be_bytes(*b"qoaf"); // :15
pub const MAX_SLICES_PER_CHANNEL_PER_FRAME: usize = 256; // :16
pub const QOA_SLICES_PER_FRAME: usize = 256; // :17
pub const QOA_FRAME_LEN: usize = QOA_SLICES_PER_FRAME * QOA_SLICE_LEN; // :18
pub const QOA_MAX_CHANNELS: usize = 8; // :19

mod ffi {
    use super::*;
    #[repr(C)]
    #[derive(Clone, Copy)]
    pub struct qoa_desc {
        pub channels: u8,
        pub sample_rate: u32,
        pub samples: u32,
    }
    #[repr(C)]
    #[derive(Default, Clone, Copy)]
    pub struct qoa_frame_header {
        pub num_channels: u8,
        pub sample_rate: u32,
        pub num_samples_per_channel: u16,
    }
    #[repr(C)]
    #[derive(Default)]
    pub struct qoa_mode {
        pub kind: c_int,
        pub channels: u8,
        pub sample_rate: u32,
        pub samples: u32,
    }
    #[repr(C)]
    #[derive(Default)]
    pub struct qoa_item {
        pub kind: c_int,
        pub sample: i16,
        pub header: qoa_frame_header,
        pub error: c_int,
    }
    #[repr(C)]
    #[derive(Clone, Copy)]
    pub struct qoa_clip {
        pub pcm_offset: u64,
        pub out_offset: u64,
        pub desc: qoa_desc,
    }
    #[repr(C)]
    #[derive(Default, Clone, Copy)]
    pub struct qoa_launch_shape {
        pub struct_bytes: u32,
        pub wavefronts: u32,
        pub chains_per_wavefront: u32,
        pub lds_table_form: u32,
        pub helper_wavefronts: u32,
        pub launches: u32,
    }
    #[repr(C)]
    #[derive(Default, Clone, Copy)]
    pub struct qoa_wave_stats {
        pub struct_bytes: u32,
        pub wavefronts: u32,
        pub slices: [u64; 8],
        pub issue_slots: [u32; 8],
        pub cycles_per_issue: f64,
        pub loop_overhead: f64,
        pub bound_cycles_per_slice: f64,
        pub cycles_per_slice_min: f64,
        pub cycles_per_slice_p50: f64,
        pub cycles_per_slice_max: f64,
        pub slowest_wavefront: u32,
        pub slowest_channels: u32,
        pub slowest_slices_of_chain: u32,
        pub slowest_slices: [u32; 8],
        pub slowest_predicted_cycles_per_slice: f64,
    }
    #[repr(C)]
    #[derive(Default)]
    pub struct qoa_stream_info {
        pub mode: qoa_mode,
        pub terminal: c_int,
        pub in_new: c_int,
        pub n_frames: u64,
        pub n_samples: u64,
    }
    pub enum qoa_encoder {}
    pub enum qoa_decoder {}
    pub enum qoa_multi {}
    pub enum qoa_encode_plan {}
    pub enum qoa_decode_plan {}
    pub enum qoa_host_pipeline {}
    /// `int64_t (*)(void *ctx, uint8_t *buf, size_t len)`: bytes stored, 0 at the end of the stream, negative on error
    pub type qoa_read_fn = unsafe extern "C" fn(*mut c_void, *mut u8, usize) -> i64;
    extern "C" {
        pub fn qoa_encode_strerror(code: c_int) -> *const c_char;
        pub fn qoa_decode_strerror(code: c_int) -> *const c_char;
        pub fn qoa_last_error_detail() -> *const c_char;
        pub fn qoa_version() -> *const c_char;
        pub fn qoa_encoded_size(channels: u32, samples: u32) -> usize;
        pub fn qoa_frame_size(channels: u32, slices: u32) -> u16;
        pub fn qoa_encoder_new(desc: *const qoa_desc, out: *mut *mut qoa_encoder) -> c_int;
        pub fn qoa_encoder_free(enc: *mut qoa_encoder);
        pub fn qoa_encoder_encode(enc: *mut qoa_encoder, pcm: *const i16, n_i16: usize, out: *mut u8, cap: usize,
                                  written: *mut usize) -> c_int;
        pub fn qoa_encoder_write_header(enc: *const qoa_encoder, out: *mut u8) -> c_int;
        pub fn qoa_encoder_encode_frame(enc: *mut qoa_encoder, pcm: *const i16, n_i16: usize, out: *mut u8,
                                        cap: usize, written: *mut usize, frame_len: *mut usize) -> c_int;
        pub fn qoa_encoder_get_lms(enc: *mut qoa_encoder, state: *mut i32) -> c_int;
        pub fn qoa_encoder_set_lms(enc: *mut qoa_encoder, state: *const i32) -> c_int;
        pub fn qoa_encode_all(pcm: *const i16, n_i16: usize, desc: *const qoa_desc, out: *mut u8, cap: usize,
                              written: *mut usize) -> c_int;
        // batches on device-resident buffers: the entry points through which a host reaches the kernels' full rate
        pub fn qoa_encode_plan_create(clips: *const qoa_clip, n_clips: usize, flags: u32, out: *mut *mut qoa_encode_plan) -> c_int;
        pub fn qoa_encode_plan_free(plan: *mut qoa_encode_plan);
        pub fn qoa_encode_plan_bytes(plan: *const qoa_encode_plan, pcm_bytes: *mut u64, out_bytes: *mut u64,
                                     samples_i16: *mut u64) -> c_int;
        pub fn qoa_encode_plan_launch_info(plan: *const qoa_encode_plan, wavefronts: *mut u32, chains_per_wavefront: *mut u32,
                                           lds_table_form: *mut u32) -> c_int;
        pub fn qoa_encode_plan_launch_shape(plan: *const qoa_encode_plan, shape: *mut qoa_launch_shape) -> c_int;
        pub fn qoa_encode_plan_run(plan: *mut qoa_encode_plan, d_pcm: *const i16, d_out: *mut u8, stream: *mut c_void) -> c_int;
        pub fn qoa_encode_plan_stats(plan: *mut qoa_encode_plan, declined_slices: *mut u64, unpacked_slices: *mut u64,
                                     speculation_misses: *mut u64) -> c_int;
        pub fn qoa_encode_plan_counters(plan: *mut qoa_encode_plan, out: *mut u64, n: usize) -> c_int;
        pub fn qoa_encode_plan_wave_stats(plan: *mut qoa_encode_plan, out: *mut qoa_wave_stats) -> c_int;
        pub fn qoa_decode_plan_from_streams(streams: *const *const u8, len: *const usize, in_offset: *const u64,
                                            out_offset: *const u64, n_streams: usize, status: *mut c_int,
                                            samples_out: *mut u64, out: *mut *mut qoa_decode_plan) -> c_int;
        pub fn qoa_decode_plan_from_clips(clips: *const qoa_clip, n_clips: usize, out: *mut *mut qoa_decode_plan) -> c_int;
        pub fn qoa_decode_plan_free(plan: *mut qoa_decode_plan);
        pub fn qoa_decode_plan_bytes(plan: *const qoa_decode_plan, in_bytes: *mut u64, pcm_bytes: *mut u64,
                                     samples_i16: *mut u64) -> c_int;
        pub fn qoa_decode_plan_run(plan: *mut qoa_decode_plan, d_in: *const u8, d_out: *mut i16, stream: *mut c_void) -> c_int;
        pub fn qoa_decode_plan_mismatches(plan: *mut qoa_decode_plan, count: *mut u32) -> c_int;
        // batches on host-resident buffers: the time-sliced pipeline
        pub fn qoa_encode_batch_host(clips: *const qoa_clip, n_clips: usize, h_pcm: *const i16, h_out: *mut u8,
                                     chunk_frames: u32) -> c_int;
        pub fn qoa_decode_batch_host(clips: *const qoa_clip, n_clips: usize, h_enc: *const u8, h_pcm: *mut i16,
                                     chunk_frames: u32) -> c_int;
        pub fn qoa_host_pipeline_create(out: *mut *mut qoa_host_pipeline) -> c_int;
        pub fn qoa_host_pipeline_free(p: *mut qoa_host_pipeline);
        pub fn qoa_host_pipeline_encode(p: *mut qoa_host_pipeline, clips: *const qoa_clip, n_clips: usize, h_pcm: *const i16,
                                        h_out: *mut u8, chunk_frames: u32) -> c_int;
        pub fn qoa_host_pipeline_decode(p: *mut qoa_host_pipeline, clips: *const qoa_clip, n_clips: usize, h_enc: *const u8,
                                        h_pcm: *mut i16, chunk_frames: u32) -> c_int;
        pub fn qoa_stream_scan(bytes: *const u8, len: usize, info: *mut qoa_stream_info) -> c_int;
        pub fn qoa_decoder_new(bytes: *const u8, len: usize, out: *mut *mut qoa_decoder) -> c_int;
        pub fn qoa_decoder_new_reader(read: qoa_read_fn, ctx: *mut c_void, out: *mut *mut qoa_decoder) -> c_int;
        pub fn qoa_decoder_new_streaming(out: *mut *mut qoa_decoder) -> c_int;
        pub fn qoa_decoder_free(dec: *mut qoa_decoder);
        pub fn qoa_decoder_mode(dec: *const qoa_decoder, mode: *mut qoa_mode) -> c_int;
        pub fn qoa_decoder_current_frame_header(dec: *const qoa_decoder, h: *mut qoa_frame_header) -> c_int;
        pub fn qoa_decoder_total_duration(dec: *const qoa_decoder) -> c_double;
        pub fn qoa_decoder_next(dec: *mut qoa_decoder, item: *mut qoa_item) -> c_int;
        pub fn qoa_decoder_read_samples(dec: *mut qoa_decoder, out: *mut i16, cap: usize) -> usize;
        pub fn qoa_decoder_decode_frame(dec: *mut qoa_decoder, frame: *const u8, len: usize, out: *mut i16,
                                        cap: usize, n_out: *mut usize) -> c_int;
        pub fn qoa_decode_all(bytes: *const u8, len: usize, channels: *mut u8, sample_rate: *mut u32,
                              out: *mut i16, cap: usize, n_out: *mut usize) -> c_int;
        // multi-GPU batches in one process (no counterpart in the reference; include/qoaudio_amd.h)
        pub fn qoa_multi_create(devices: *const c_int, n_devices: c_int, out: *mut *mut qoa_multi) -> c_int;
        pub fn qoa_multi_create_flags(devices: *const c_int, n_devices: c_int, flags: u32, out: *mut *mut qoa_multi) -> c_int;
        pub fn qoa_multi_free(m: *mut qoa_multi);
        pub fn qoa_multi_shard(n_clips: usize, n_devices: c_int, index: c_int, first: *mut usize, count: *mut usize) -> c_int;
        pub fn qoa_multi_encode(m: *mut qoa_multi, clips: *const qoa_clip, n_clips: usize, pcm: *const *const i16,
                                out: *mut u8, chunk_frames: u32) -> c_int;
        pub fn qoa_multi_decode(m: *mut qoa_multi, clips: *const qoa_clip, n_clips: usize, enc: *const u8,
                                pcm: *const *mut i16, chunk_frames: u32) -> c_int;
        pub fn qoa_encode_batch_multi(devices: *const c_int, n_devices: c_int, clips: *const qoa_clip, n_clips: usize,
                                      pcm: *const *const i16, out: *mut u8, chunk_frames: u32) -> c_int;
        pub fn qoa_decode_batch_multi(devices: *const c_int, n_devices: c_int, clips: *const qoa_clip, n_clips: usize,
                                      enc: *const u8, pcm: *const *mut i16, chunk_frames: u32) -> c_int;
    }
}

fn io_detail(kind: io::ErrorKind) -> io::Error {
    let s = unsafe { CStr::from_ptr(ffi::qoa_last_error_detail()) }.to_string_lossy().into_owned();
    io::Error::new(kind, s)
}

/// The decoding mode of the QOA file. (:30-45)
#[derive(Debug, Clone, PartialEq)]
pub enum ProcessingMode {
    FixedSamples { channels: u8, sample_rate: u32, samples: u32 },
    Streaming,
}

/// Description of QOA file properties for encoding (:94-102)
#[derive(Debug, Clone)]
pub struct QoaDesc {
    pub channels: u8,
    pub sample_rate: u32,
    pub samples: u32,
}

/// Errors that can occur during QOA encoding (:105-115)
#[derive(Debug)]
pub enum EncodeError {
    InvalidChannels,
    InvalidSampleRate,
    InvalidSamples,
    IoError(io::Error),
}

fn enc_err(code: c_int) -> EncodeError {
    match code {
        1 => EncodeError::InvalidChannels,
        2 => EncodeError::InvalidSampleRate,
        3 => EncodeError::InvalidSamples,
        _ => EncodeError::IoError(io_detail(io::ErrorKind::Other)),
    }
}

#[derive(Debug)]
pub enum DecodeError {
    NotQoaFile,
    NoSamples,
    InvalidFrameHeader,
    IncompatibleFrame,
    IoError(io::Error),
}

fn dec_err(code: c_int) -> DecodeError {
    match code {
        1 => DecodeError::NotQoaFile,
        2 => DecodeError::NoSamples,
        3 => DecodeError::InvalidFrameHeader,
        4 => DecodeError::IncompatibleFrame,
        _ => DecodeError::IoError(io_detail(io::ErrorKind::UnexpectedEof)),
    }
}

impl std::error::Error for DecodeError {}
impl Display for DecodeError { // :877-887
    fn fmt(&self, f: &mut fmt::Formatter) -> fmt::Result {
        match self {
            DecodeError::NotQoaFile => write!(f, "File is not a qoa file"),
            DecodeError::NoSamples => write!(f, "File has no samples"),
            DecodeError::InvalidFrameHeader => write!(f, "File has invalid frame header"),
            DecodeError::IncompatibleFrame => write!(f, "Incompatible frame header"),
            DecodeError::IoError(e) => write!(f, "IO error: {}", e),
        }
    }
}
impl From<io::Error> for DecodeError {
    fn from(inner: io::Error) -> Self { DecodeError::IoError(inner) }
}
impl std::error::Error for EncodeError {}
impl Display for EncodeError { // :897-906
    fn fmt(&self, f: &mut fmt::Formatter) -> fmt::Result {
        match self {
            EncodeError::InvalidChannels => write!(f, "Invalid number of channels (must be 1-8)"),
            EncodeError::InvalidSampleRate => write!(f, "Invalid sample rate (must be > 0)"),
            EncodeError::InvalidSamples => write!(f, "Invalid number of samples (must be > 0)"),
            EncodeError::IoError(e) => write!(f, "IO error: {}", e),
        }
    }
}
impl From<io::Error> for EncodeError {
    fn from(inner: io::Error) -> Self { EncodeError::IoError(inner) }
}

/// The metadata at the beginning of each frame of slices. (:772-781)
#[derive(Debug, Copy, Clone, Default)]
pub struct FrameHeader {
    pub num_channels: u8,
    pub sample_rate: u32,
    pub num_samples_per_channel: u16,
}

/// Return type of `QoaDecoder::next`. (:654-659)
#[derive(Debug)]
pub enum QoaItem {
    Sample(i16),
    FrameHeader(FrameHeader),
}

/// A fully decoded QOA file. (:695-704)
pub struct DecodedQoa {
    pub num_channels: u8,
    pub sample_rate: u32,
    pub samples: Vec<i16>,
}

/// Encoder for QOA files (:76-91).  LMS state lives in device memory inside the handle.
#[derive(Debug)]
pub struct QoaEncoder {
    h: *mut ffi::qoa_encoder,
    channels: u8,
    samples: u32,
}
unsafe impl Send for QoaEncoder {}

impl QoaEncoder {
    pub fn new(desc: &QoaDesc) -> Result<Self, EncodeError> { // :335-361
        let d = ffi::qoa_desc { channels: desc.channels, sample_rate: desc.sample_rate, samples: desc.samples };
        let mut h = std::ptr::null_mut();
        match unsafe { ffi::qoa_encoder_new(&d, &mut h) } {
            0 => Ok(Self { h, channels: desc.channels, samples: desc.samples }),
            c => Err(enc_err(c)),
        }
    }
    pub fn encode(&mut self, sample_data: &[i16]) -> Result<Vec<u8>, EncodeError> { // :367-398
        let mut buf = vec![0u8; unsafe { ffi::qoa_encoded_size(self.channels as u32, self.samples) }];
        let mut written = 0usize;
        let rc = unsafe {
            ffi::qoa_encoder_encode(self.h, sample_data.as_ptr(), sample_data.len(), buf.as_mut_ptr(), buf.len(),
                                    &mut written)
        };
        if rc != 0 { return Err(enc_err(rc)); }
        buf.truncate(written);
        Ok(buf)
    }
    pub fn write_header<W: io::Write>(&self, writer: &mut W) -> Result<(), EncodeError> { // :404-408
        let mut b = [0u8; 8];
        let rc = unsafe { ffi::qoa_encoder_write_header(self.h, b.as_mut_ptr()) };
        if rc != 0 { return Err(enc_err(rc)); }
        writer.write_all(&b)?;
        Ok(())
    }
    pub fn encode_frame<W: io::Write>(&mut self, sample_data: &[i16], writer: &mut W) -> Result<usize, EncodeError> { // :418-439
        let mut buf = vec![0u8; 8 + 16 * QOA_MAX_CHANNELS + 8 * QOA_SLICES_PER_FRAME * QOA_MAX_CHANNELS];
        let (mut written, mut frame_len) = (0usize, 0usize);
        let rc = unsafe {
            ffi::qoa_encoder_encode_frame(self.h, sample_data.as_ptr(), sample_data.len(), buf.as_mut_ptr(), buf.len(),
                                          &mut written, &mut frame_len)
        };
        if rc != 0 { return Err(enc_err(rc)); }
        writer.write_all(&buf[..written])?;
        Ok(frame_len)
    }
}
impl Drop for QoaEncoder {
    fn drop(&mut self) { unsafe { ffi::qoa_encoder_free(self.h) } }
}

/// Encode PCM audio data to QOA format. (:756-763)
pub fn encode_all(sample_data: &[i16], desc: &QoaDesc) -> Result<Vec<u8>, EncodeError> {
    let mut encoder = QoaEncoder::new(desc)?;
    encoder.encode(sample_data)
}

/// Decoder of QOA files (:47-68).  The reader is pulled through a C callback, one frame at a time, as `next` runs out of
/// samples (`qoa_decoder_new_reader`); the GPU decodes whole frames and `next` replays the reference's item order.  `mode`
/// and `current_frame` mirror the reference's fields of the same purpose (:60, :63) so that the accessors can hand out
/// references.
#[derive(Debug)]
pub struct QoaDecoder<R> {
    h: *mut ffi::qoa_decoder,
    cell: Option<Box<ReaderCell<R>>>, // boxed: the C side keeps its address as the callback's context
    mode: ProcessingMode,
    current_frame: FrameHeader,
}
unsafe impl<R: Send> Send for QoaDecoder<R> {}

/// The reader and the last error it raised (handed to the caller instead of a generic IoError).
#[derive(Debug)]
struct ReaderCell<R> {
    reader: R,
    error: Option<io::Error>,
}

/// `qoa_read_fn` over any `io::Read`: bytes stored, 0 at the end, -1 after an error (kept in the cell).
unsafe extern "C" fn read_trampoline<R: io::Read>(ctx: *mut c_void, buf: *mut u8, len: usize) -> i64 {
    let cell = &mut *(ctx as *mut ReaderCell<R>);
    let dst = std::slice::from_raw_parts_mut(buf, len);
    loop {
        match cell.reader.read(dst) {
            Ok(n) => return n as i64,
            Err(e) if e.kind() == io::ErrorKind::Interrupted => continue, // as read_exact does
            Err(e) => {
                cell.error = Some(e);
                return -1;
            }
        }
    }
}

fn query_mode(h: *mut ffi::qoa_decoder) -> ProcessingMode {
    let mut m = ffi::qoa_mode::default();
    unsafe { ffi::qoa_decoder_mode(h, &mut m) };
    if m.kind == 1 {
        ProcessingMode::Streaming
    } else {
        ProcessingMode::FixedSamples { channels: m.channels, sample_rate: m.sample_rate, samples: m.samples }
    }
}

fn query_header(h: *mut ffi::qoa_decoder) -> FrameHeader {
    let mut fh = ffi::qoa_frame_header::default();
    unsafe { ffi::qoa_decoder_current_frame_header(h, &mut fh) };
    FrameHeader { num_channels: fh.num_channels, sample_rate: fh.sample_rate, num_samples_per_channel: fh.num_samples_per_channel }
}

impl<R> QoaDecoder<R> {
    /// IoError: the reader's own error when it raised one, else the library's detail string
    fn map_err(&mut self, code: c_int) -> DecodeError {
        if code == 5 {
            if let Some(e) = self.cell.as_mut().and_then(|c| c.error.take()) {
                return DecodeError::IoError(e);
            }
        }
        dec_err(code)
    }
}

impl<R: io::Read> QoaDecoder<R> {
    pub fn new(mut reader: R) -> Result<Self, DecodeError> { // :127-169
        let _ = &mut reader;
        let mut cell = Box::new(ReaderCell { reader, error: None });
        let mut h = std::ptr::null_mut();
        let ctx = &mut *cell as *mut ReaderCell<R> as *mut c_void;
        match unsafe { ffi::qoa_decoder_new_reader(read_trampoline::<R>, ctx, &mut h) } {
            0 => Ok(Self { h, cell: Some(cell), mode: query_mode(h), current_frame: query_header(h) }),
            5 if cell.error.is_some() => Err(DecodeError::IoError(cell.error.take().unwrap())),
            c => Err(dec_err(c)),
        }
    }
    pub fn into_inner(mut self) -> R { // :172-174 -- the decoder goes first: it holds the cell's address
        unsafe { ffi::qoa_decoder_free(self.h) };
        self.h = std::ptr::null_mut();
        self.cell.take().unwrap().reader
    }
    pub fn mode(&self) -> &ProcessingMode { &self.mode } // :180-182
    pub fn current_frame_header(&self) -> &FrameHeader { &self.current_frame } // :186-188
    pub fn total_duration(&self) -> Option<Duration> { // :190-201
        let d = unsafe { ffi::qoa_decoder_total_duration(self.h) };
        if d < 0.0 { None } else { Some(Duration::from_secs_f64(d)) }
    }
}

impl QoaDecoder<io::BufReader<File>> {
    pub fn open<P: AsRef<Path>>(path: P) -> Result<QoaDecoder<io::BufReader<File>>, DecodeError> { // :619-625
        let file = File::open(path)?;
        QoaDecoder::new(io::BufReader::new(file))
    }
}

impl QoaDecoder<Cursor<Vec<u8>>> {
    pub fn new_streaming() -> Result<Self, DecodeError> { // :632-638
        let mut h = std::ptr::null_mut();
        match unsafe { ffi::qoa_decoder_new_streaming(&mut h) } {
            0 => Ok(Self { h, cell: Some(Box::new(ReaderCell { reader: Cursor::new(Vec::new()), error: None })),
                           mode: ProcessingMode::Streaming, current_frame: FrameHeader::default() }),
            c => Err(dec_err(c)),
        }
    }
    pub fn decode_frame(&mut self, frame_data: &[u8]) -> Result<Vec<i16>, DecodeError> { // :641-651
        let mut n = 0usize;
        let rc = unsafe {
            ffi::qoa_decoder_decode_frame(self.h, frame_data.as_ptr(), frame_data.len(), std::ptr::null_mut(), 0, &mut n)
        };
        self.current_frame = query_header(self.h); // the frames just consumed moved it (:246-289)
        if n == 0 {
            return if rc != 0 { Err(dec_err(rc)) } else { Ok(Vec::new()) };
        }
        let mut out = vec![0i16; n];
        let rc = unsafe { ffi::qoa_decoder_decode_frame(self.h, std::ptr::null(), 0, out.as_mut_ptr(), n, &mut n) };
        if rc != 0 { return Err(dec_err(rc)); }
        Ok(out)
    }
}

impl<R: io::Read> Iterator for QoaDecoder<R> { // :661-693
    type Item = Result<QoaItem, DecodeError>;
    fn next(&mut self) -> Option<Self::Item> {
        let mut it = ffi::qoa_item::default();
        unsafe { ffi::qoa_decoder_next(self.h, &mut it) };
        match it.kind {
            0 => Some(Ok(QoaItem::Sample(it.sample))),
            1 => {
                // the reference updates current_frame.header in the call that parses the header (:246-289, :683-686)
                self.current_frame = FrameHeader {
                    num_channels: it.header.num_channels,
                    sample_rate: it.header.sample_rate,
                    num_samples_per_channel: it.header.num_samples_per_channel,
                };
                Some(Ok(QoaItem::FrameHeader(self.current_frame)))
            }
            2 => None,
            _ => Some(Err(self.map_err(it.error))),
        }
    }
}
impl<R> Drop for QoaDecoder<R> {
    fn drop(&mut self) {
        if !self.h.is_null() {
            unsafe { ffi::qoa_decoder_free(self.h) } // before `cell` is dropped: the handle holds its address
        }
    }
}

/// Decode all samples of a QOA file read from `reader`. (:706-747)
pub fn decode_all<R: io::Read>(mut reader: R) -> Result<DecodedQoa, DecodeError> {
    let mut bytes = Vec::new();
    reader.read_to_end(&mut bytes)?;
    let (mut ch, mut sr, mut n) = (0u8, 0u32, 0usize);
    let rc = unsafe { ffi::qoa_decode_all(bytes.as_ptr(), bytes.len(), &mut ch, &mut sr, std::ptr::null_mut(), 0, &mut n) };
    if rc != 0 { return Err(dec_err(rc)); }
    let mut samples = vec![0i16; n];
    if n > 0 {
        let rc = unsafe { ffi::qoa_decode_all(bytes.as_ptr(), bytes.len(), &mut ch, &mut sr, samples.as_mut_ptr(), n, &mut n) };
        if rc != 0 { return Err(dec_err(rc)); }
    }
    Ok(DecodedQoa { num_channels: ch, sample_rate: sr, samples })
}

/// Same as `decode_all` but open the file and wrap in a BufReader first. (:749-754)
pub fn open_and_decode_all<P: AsRef<Path>>(path: P) -> Result<DecodedQoa, DecodeError> {
    let file = File::open(path.as_ref())?;
    decode_all(io::BufReader::new(file))
}

/// One clip of a batch (`qoa_clip`): `pcm_offset` counts i16 elements from its shard's PCM base, `out_offset` bytes from the
/// batch's output base.
#[derive(Debug, Clone)]
pub struct BatchClip {
    pub pcm_offset: u64,
    pub out_offset: u64,
    pub desc: QoaDesc,
}

/// Several GPUs driven from one process (`qoa_multi_*`; no counterpart in the reference, which is single-threaded: per
/// clip this is `encode_all`, :760-763).  Device `i` encodes shard `i` of the block partition `shard()` describes.
pub struct MultiGpu {
    h: *mut ffi::qoa_multi,
    n_devices: usize,
}
unsafe impl Send for MultiGpu {}

impl MultiGpu {
    pub fn new(devices: &[i32]) -> Result<Self, EncodeError> {
        let mut h = std::ptr::null_mut();
        match unsafe { ffi::qoa_multi_create(devices.as_ptr(), devices.len() as c_int, &mut h) } {
            0 => Ok(Self { h, n_devices: devices.len() }),
            c => Err(enc_err(c)),
        }
    }
    /// `(first, count)` of shard `index`
    pub fn shard(&self, n_clips: usize, index: usize) -> (usize, usize) {
        let (mut first, mut count) = (0usize, 0usize);
        unsafe { ffi::qoa_multi_shard(n_clips, self.n_devices as c_int, index as c_int, &mut first, &mut count) };
        (first, count)
    }
    /// # Safety
    /// `pcm[i]` must be valid for every clip of shard `i` (host memory, or memory of device `i`), `out` for every clip's
    /// `qoa_encoded_size` bytes at its `out_offset` (host memory, or memory of the first device).
    pub unsafe fn encode(&mut self, clips: &[BatchClip], pcm: &[*const i16], out: *mut u8, chunk_frames: u32)
                         -> Result<(), EncodeError> {
        if pcm.len() != self.n_devices { return Err(EncodeError::InvalidSamples); }
        let c: Vec<ffi::qoa_clip> = clips.iter().map(|c| ffi::qoa_clip {
            pcm_offset: c.pcm_offset, out_offset: c.out_offset,
            desc: ffi::qoa_desc { channels: c.desc.channels, sample_rate: c.desc.sample_rate, samples: c.desc.samples } }).collect();
        match ffi::qoa_multi_encode(self.h, c.as_ptr(), c.len(), pcm.as_ptr(), out, chunk_frames) {
            0 => Ok(()),
            e => Err(enc_err(e)),
        }
    }
    /// # Safety
    /// as `encode`, with the roles of `enc` (one source for the batch) and `pcm[i]` (destination of shard `i`) swapped.
    pub unsafe fn decode(&mut self, clips: &[BatchClip], enc: *const u8, pcm: &[*mut i16], chunk_frames: u32)
                         -> Result<(), DecodeError> {
        if pcm.len() != self.n_devices { return Err(DecodeError::InvalidFrameHeader); }
        let c: Vec<ffi::qoa_clip> = clips.iter().map(|c| ffi::qoa_clip {
            pcm_offset: c.pcm_offset, out_offset: c.out_offset,
            desc: ffi::qoa_desc { channels: c.desc.channels, sample_rate: c.desc.sample_rate, samples: c.desc.samples } }).collect();
        match ffi::qoa_multi_decode(self.h, c.as_ptr(), c.len(), enc, pcm.as_ptr(), chunk_frames) {
            0 => Ok(()),
            e => Err(dec_err(e)),
        }
    }
}
impl Drop for MultiGpu {
    fn drop(&mut self) { unsafe { ffi::qoa_multi_free(self.h) } }
}

// ------------------------------------------------------------------------------------------------------------------------
// Batches.  No counterpart in the reference (per clip this is `encode_all`, :760-763, and `decode_all`, :713-747): the
// entry points through which a Rust host reaches the kernels' rate -- `EncodePlan` / `DecodePlan` on buffers that already
// sit in device memory (42 G samples/s encode, 2 T samples/s decode on one MI355X), `HostPipeline` from / to host memory
// (PCIe-bound, ~27 G samples/s).  Flags: the `QOA_PLAN_*` constants of include/qoaudio_amd.h.
// ------------------------------------------------------------------------------------------------------------------------
pub const QOA_PLAN_DEFAULT: u32 = 0;
pub const QOA_PLAN_COUNTERS: u32 = 512;
pub const QOA_PLAN_WAVE_STATS: u32 = 32768;

fn ffi_clips(clips: &[BatchClip]) -> Vec<ffi::qoa_clip> {
    clips.iter().map(|c| ffi::qoa_clip {
        pcm_offset: c.pcm_offset, out_offset: c.out_offset,
        desc: ffi::qoa_desc { channels: c.desc.channels, sample_rate: c.desc.sample_rate, samples: c.desc.samples } }).collect()
}

/// do all clips lie inside a PCM buffer of `n_i16` samples and an encoded buffer of `n_bytes` bytes?
fn clips_fit(clips: &[BatchClip], n_i16: usize, n_bytes: usize) -> bool {
    clips.iter().all(|c| {
        let pcm_end = c.pcm_offset as u128 + c.desc.samples as u128 * c.desc.channels as u128;
        let enc_end = c.out_offset as u128 + encoded_size(c.desc.channels, c.desc.samples) as u128;
        pcm_end <= n_i16 as u128 && enc_end <= n_bytes as u128
    })
}

/// `qoa_encoded_size` (:374-378)
pub fn encoded_size(channels: u8, samples: u32) -> usize { unsafe { ffi::qoa_encoded_size(channels as u32, samples) } }
/// `qoa_frame_size` (:602-604)
pub fn frame_size(channels: u32, slices: u32) -> u16 { unsafe { ffi::qoa_frame_size(channels, slices) } }
/// "qoaudio_amd <version> gfx950"
pub fn version() -> String { unsafe { CStr::from_ptr(ffi::qoa_version()) }.to_string_lossy().into_owned() }
/// The library's own Display strings of the two error enums (the `Display` impls above carry the same text).
pub fn encode_strerror(code: i32) -> String { unsafe { CStr::from_ptr(ffi::qoa_encode_strerror(code)) }.to_string_lossy().into_owned() }
pub fn decode_strerror(code: i32) -> String { unsafe { CStr::from_ptr(ffi::qoa_decode_strerror(code)) }.to_string_lossy().into_owned() }

/// `encode_all` through the library's one-call entry point (`qoa_encode_all`): same result as `encode_all` above.
pub fn encode_all_oneshot(sample_data: &[i16], desc: &QoaDesc) -> Result<Vec<u8>, EncodeError> {
    let d = ffi::qoa_desc { channels: desc.channels, sample_rate: desc.sample_rate, samples: desc.samples };
    let mut buf = vec![0u8; if (1..=8).contains(&desc.channels) { encoded_size(desc.channels, desc.samples) } else { 16 }];
    let mut written = 0usize;
    match unsafe { ffi::qoa_encode_all(sample_data.as_ptr(), sample_data.len(), &d, buf.as_mut_ptr(), buf.len(), &mut written) } {
        0 => { buf.truncate(written); Ok(buf) }
        c => Err(enc_err(c)),
    }
}

impl QoaEncoder {
    /// Carried LMS state, `[history[4], weights[4]]` per channel as full i32 (:70-74, :89): checkpoint / resume.
    pub fn lms_state(&mut self) -> Result<Vec<[i32; 8]>, EncodeError> {
        let mut flat = vec![0i32; self.channels as usize * 8];
        match unsafe { ffi::qoa_encoder_get_lms(self.h, flat.as_mut_ptr()) } {
            0 => Ok(flat.chunks_exact(8).map(|c| [c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7]]).collect()),
            c => Err(enc_err(c)),
        }
    }
    pub fn set_lms_state(&mut self, state: &[[i32; 8]]) -> Result<(), EncodeError> {
        if state.len() != self.channels as usize { return Err(EncodeError::InvalidChannels); }
        let flat: Vec<i32> = state.iter().flatten().copied().collect();
        match unsafe { ffi::qoa_encoder_set_lms(self.h, flat.as_ptr()) } { 0 => Ok(()), c => Err(enc_err(c)) }
    }
}

impl<R: io::Read> QoaDecoder<R> {
    /// Bulk form of `next()`: consecutive `Sample` items copied into `out`; stops early at the next frame header, the end or
    /// an error (which the following `next()` returns).  Returns the count.
    pub fn read_samples(&mut self, out: &mut [i16]) -> usize {
        unsafe { ffi::qoa_decoder_read_samples(self.h, out.as_mut_ptr(), out.len()) }
    }
}

/// `qoa_launch_shape`: how a plan's launches are shaped.
#[derive(Debug, Clone, Copy, Default)]
pub struct LaunchShape {
    pub wavefronts: u32,
    pub chains_per_wavefront: u32,
    pub lds_table_form: bool,
    pub helper_wavefronts: bool,
    pub launches: u32,
}

/// `qoa_wave_stats`: what the last run of an `EncodePlan` created with `QOA_PLAN_WAVE_STATS` cost, wavefront by wavefront
/// (include/qoaudio_amd.h explains the forms and the bound: clipped material runs at most ~1.6x slower than music).
#[derive(Debug, Clone, Copy, Default)]
pub struct WaveStats {
    pub wavefronts: u32,
    pub slices: [u64; 8],
    pub issue_slots: [u32; 8],
    pub cycles_per_issue: f64,
    pub loop_overhead: f64,
    pub bound_cycles_per_slice: f64,
    pub cycles_per_slice_min: f64,
    pub cycles_per_slice_p50: f64,
    pub cycles_per_slice_max: f64,
    pub slowest_wavefront: u32,
    pub slowest_channels: u32,
    pub slowest_slices_of_chain: u32,
    pub slowest_wavefront_form_mix: [u32; 8],
    pub slowest_predicted_cycles_per_slice: f64,
}

/// Chain table of a batch of clips, resident on the calling thread's current HIP device (`qoa_encode_plan_*`).
pub struct EncodePlan {
    h: *mut ffi::qoa_encode_plan,
}
unsafe impl Send for EncodePlan {}

impl EncodePlan {
    pub fn new(clips: &[BatchClip], flags: u32) -> Result<Self, EncodeError> {
        let c = ffi_clips(clips);
        let mut h = std::ptr::null_mut();
        match unsafe { ffi::qoa_encode_plan_create(c.as_ptr(), c.len(), flags, &mut h) } {
            0 => Ok(Self { h }),
            e => Err(enc_err(e)),
        }
    }
    /// `(pcm bytes read, encoded bytes written, i16 samples)` of one run
    pub fn bytes(&self) -> (u64, u64, u64) {
        let (mut a, mut b, mut c) = (0u64, 0u64, 0u64);
        unsafe { ffi::qoa_encode_plan_bytes(self.h, &mut a, &mut b, &mut c) };
        (a, b, c)
    }
    pub fn launch_shape(&self) -> LaunchShape {
        let mut s = ffi::qoa_launch_shape { struct_bytes: std::mem::size_of::<ffi::qoa_launch_shape>() as u32, ..Default::default() };
        unsafe { ffi::qoa_encode_plan_launch_shape(self.h, &mut s) };
        LaunchShape { wavefronts: s.wavefronts, chains_per_wavefront: s.chains_per_wavefront, lds_table_form: s.lds_table_form != 0,
                      helper_wavefronts: s.helper_wavefronts != 0, launches: s.launches }
    }
    /// `(wavefronts, chains per wavefront, LDS-table form)`: the older three-value form of `launch_shape`
    pub fn launch_info(&self) -> (u32, u32, bool) {
        let (mut a, mut b, mut c) = (0u32, 0u32, 0u32);
        unsafe { ffi::qoa_encode_plan_launch_info(self.h, &mut a, &mut b, &mut c) };
        (a, b, c != 0)
    }
    /// Enqueue the batch encode on `stream` (a `hipStream_t`, may be null).  Asynchronous.
    /// # Safety
    /// `d_pcm` / `d_out` must be device memory valid for every clip of the plan until the stream has run the launch.
    pub unsafe fn run(&mut self, d_pcm: *const i16, d_out: *mut u8, stream: *mut c_void) -> Result<(), EncodeError> {
        match ffi::qoa_encode_plan_run(self.h, d_pcm, d_out, stream) { 0 => Ok(()), e => Err(enc_err(e)) }
    }
    /// `(declined, unpacked, speculation misses)`; plans created with `QOA_PLAN_COUNTERS` only
    pub fn stats(&mut self) -> Result<(u64, u64, u64), EncodeError> {
        let (mut a, mut b, mut c) = (0u64, 0u64, 0u64);
        match unsafe { ffi::qoa_encode_plan_stats(self.h, &mut a, &mut b, &mut c) } { 0 => Ok((a, b, c)), e => Err(enc_err(e)) }
    }
    pub fn counters(&mut self) -> Result<[u64; 10], EncodeError> {
        let mut out = [0u64; 10];
        match unsafe { ffi::qoa_encode_plan_counters(self.h, out.as_mut_ptr(), out.len()) } { 0 => Ok(out), e => Err(enc_err(e)) }
    }
    /// plans created with `QOA_PLAN_WAVE_STATS` only
    pub fn wave_stats(&mut self) -> Result<WaveStats, EncodeError> {
        let mut w = ffi::qoa_wave_stats { struct_bytes: std::mem::size_of::<ffi::qoa_wave_stats>() as u32, ..Default::default() };
        match unsafe { ffi::qoa_encode_plan_wave_stats(self.h, &mut w) } {
            0 => Ok(WaveStats { wavefronts: w.wavefronts, slices: w.slices, issue_slots: w.issue_slots, cycles_per_issue: w.cycles_per_issue,
                                loop_overhead: w.loop_overhead, bound_cycles_per_slice: w.bound_cycles_per_slice,
                                cycles_per_slice_min: w.cycles_per_slice_min, cycles_per_slice_p50: w.cycles_per_slice_p50,
                                cycles_per_slice_max: w.cycles_per_slice_max, slowest_wavefront: w.slowest_wavefront,
                                slowest_channels: w.slowest_channels, slowest_slices_of_chain: w.slowest_slices_of_chain,
                                slowest_wavefront_form_mix: w.slowest_slices,
                                slowest_predicted_cycles_per_slice: w.slowest_predicted_cycles_per_slice }),
            e => Err(enc_err(e)),
        }
    }
}
impl Drop for EncodePlan {
    fn drop(&mut self) { unsafe { ffi::qoa_encode_plan_free(self.h) } }
}

/// Frame table of a batch of encoded streams, resident on the current device (`qoa_decode_plan_*`).
pub struct DecodePlan {
    h: *mut ffi::qoa_decode_plan,
    /// `from_streams`: per stream, `Ok(samples it yields)` or the error that ends it (after that many samples)
    pub streams: Vec<(u64, Option<DecodeError>)>,
}
unsafe impl Send for DecodePlan {}

impl DecodePlan {
    /// Streams laid out as `EncodePlan::run` wrote them (`pcm_offset` = PCM destination, `out_offset` = encoded source).
    pub fn from_clips(clips: &[BatchClip]) -> Result<Self, DecodeError> {
        let c = ffi_clips(clips);
        let mut h = std::ptr::null_mut();
        match unsafe { ffi::qoa_decode_plan_from_clips(c.as_ptr(), c.len(), &mut h) } {
            0 => Ok(Self { h, streams: Vec::new() }),
            e => Err(dec_err(e)),
        }
    }
    /// Arbitrary QOA streams, parsed on the host the way the reference decoder walks them (:205-289, :320-328); the device
    /// copy of stream `i` starts at byte `in_offset[i]`, its PCM lands at i16 index `out_offset[i]`.
    pub fn from_streams(streams: &[&[u8]], in_offset: &[u64], out_offset: &[u64]) -> Result<Self, DecodeError> {
        if in_offset.len() != streams.len() || out_offset.len() != streams.len() { return Err(DecodeError::InvalidFrameHeader); }
        let ptrs: Vec<*const u8> = streams.iter().map(|s| s.as_ptr()).collect();
        let lens: Vec<usize> = streams.iter().map(|s| s.len()).collect();
        let mut status = vec![0 as c_int; streams.len()];
        let mut n_out = vec![0u64; streams.len()];
        let mut h = std::ptr::null_mut();
        let rc = unsafe {
            ffi::qoa_decode_plan_from_streams(ptrs.as_ptr(), lens.as_ptr(), in_offset.as_ptr(), out_offset.as_ptr(), streams.len(),
                                              status.as_mut_ptr(), n_out.as_mut_ptr(), &mut h)
        };
        if rc != 0 { return Err(dec_err(rc)); }
        let streams = n_out.iter().zip(&status).map(|(&n, &s)| (n, if s == 0 { None } else { Some(dec_err(s)) })).collect();
        Ok(Self { h, streams })
    }
    /// `(encoded bytes read, pcm bytes written, i16 samples)` of one run
    pub fn bytes(&self) -> (u64, u64, u64) {
        let (mut a, mut b, mut c) = (0u64, 0u64, 0u64);
        unsafe { ffi::qoa_decode_plan_bytes(self.h, &mut a, &mut b, &mut c) };
        (a, b, c)
    }
    /// # Safety
    /// `d_in` / `d_out` must be device memory valid for every stream of the plan until the stream has run the launches.
    pub unsafe fn run(&mut self, d_in: *const u8, d_out: *mut i16, stream: *mut c_void) -> Result<(), DecodeError> {
        match ffi::qoa_decode_plan_run(self.h, d_in, d_out, stream) { 0 => Ok(()), e => Err(dec_err(e)) }
    }
    /// frames whose header on the device disagreed with the plan in the last run (synchronises)
    pub fn mismatches(&mut self) -> Result<u32, DecodeError> {
        let mut n = 0u32;
        match unsafe { ffi::qoa_decode_plan_mismatches(self.h, &mut n) } { 0 => Ok(n), e => Err(dec_err(e)) }
    }
}
impl Drop for DecodePlan {
    fn drop(&mut self) { unsafe { ffi::qoa_decode_plan_free(self.h) } }
}

/// The time-sliced host pipeline as a persistent handle (`qoa_host_pipeline_*`): `encode` is `encode_all` per clip from host
/// PCM to host bytes, `decode` the way back; the slices are bounds-checked against the clip list, so both are safe.
pub struct HostPipeline {
    h: *mut ffi::qoa_host_pipeline,
}
unsafe impl Send for HostPipeline {}

impl HostPipeline {
    pub fn new() -> Result<Self, EncodeError> {
        let mut h = std::ptr::null_mut();
        match unsafe { ffi::qoa_host_pipeline_create(&mut h) } { 0 => Ok(Self { h }), e => Err(enc_err(e)) }
    }
    pub fn encode(&mut self, clips: &[BatchClip], pcm: &[i16], out: &mut [u8], chunk_frames: u32) -> Result<(), EncodeError> {
        if !clips_fit(clips, pcm.len(), out.len()) { return Err(EncodeError::InvalidSamples); }
        let c = ffi_clips(clips);
        match unsafe { ffi::qoa_host_pipeline_encode(self.h, c.as_ptr(), c.len(), pcm.as_ptr(), out.as_mut_ptr(), chunk_frames) } {
            0 => Ok(()),
            e => Err(enc_err(e)),
        }
    }
    pub fn decode(&mut self, clips: &[BatchClip], enc: &[u8], pcm: &mut [i16], chunk_frames: u32) -> Result<(), DecodeError> {
        if !clips_fit(clips, pcm.len(), enc.len()) { return Err(DecodeError::InvalidFrameHeader); }
        let c = ffi_clips(clips);
        match unsafe { ffi::qoa_host_pipeline_decode(self.h, c.as_ptr(), c.len(), enc.as_ptr(), pcm.as_mut_ptr(), chunk_frames) } {
            0 => Ok(()),
            e => Err(dec_err(e)),
        }
    }
}
impl Drop for HostPipeline {
    fn drop(&mut self) { unsafe { ffi::qoa_host_pipeline_free(self.h) } }
}

/// One-shot forms of `HostPipeline::encode` / `decode` (the handle lives for the call).
pub fn encode_batch_host(clips: &[BatchClip], pcm: &[i16], out: &mut [u8], chunk_frames: u32) -> Result<(), EncodeError> {
    if !clips_fit(clips, pcm.len(), out.len()) { return Err(EncodeError::InvalidSamples); }
    let c = ffi_clips(clips);
    match unsafe { ffi::qoa_encode_batch_host(c.as_ptr(), c.len(), pcm.as_ptr(), out.as_mut_ptr(), chunk_frames) } {
        0 => Ok(()),
        e => Err(enc_err(e)),
    }
}
pub fn decode_batch_host(clips: &[BatchClip], enc: &[u8], pcm: &mut [i16], chunk_frames: u32) -> Result<(), DecodeError> {
    if !clips_fit(clips, pcm.len(), enc.len()) { return Err(DecodeError::InvalidFrameHeader); }
    let c = ffi_clips(clips);
    match unsafe { ffi::qoa_decode_batch_host(c.as_ptr(), c.len(), enc.as_ptr(), pcm.as_mut_ptr(), chunk_frames) } {
        0 => Ok(()),
        e => Err(dec_err(e)),
    }
}

impl MultiGpu {
    /// `qoa_multi_create_flags`: `QOA_MULTI_ALLOW_DUPLICATE_DEVICES` (= 1) is a test mode for boxes with fewer GPUs than shards.
    pub fn with_flags(devices: &[i32], flags: u32) -> Result<Self, EncodeError> {
        let mut h = std::ptr::null_mut();
        match unsafe { ffi::qoa_multi_create_flags(devices.as_ptr(), devices.len() as c_int, flags, &mut h) } {
            0 => Ok(Self { h, n_devices: devices.len() }),
            c => Err(enc_err(c)),
        }
    }
}

/// `qoa_stream_info`: how the reference decoder's walk over a stream ends (host only, no GPU work).
#[derive(Debug)]
pub struct StreamInfo {
    pub mode: ProcessingMode,
    /// `None` = clean end, else the error `QoaDecoder::new` or the iterator raises
    pub terminal: Option<DecodeError>,
    /// the error is raised by `QoaDecoder::new` itself
    pub in_new: bool,
    pub n_frames: u64,
    pub n_samples: u64,
}
pub fn stream_scan(bytes: &[u8]) -> StreamInfo {
    let mut i = ffi::qoa_stream_info::default();
    unsafe { ffi::qoa_stream_scan(bytes.as_ptr(), bytes.len(), &mut i) };
    let mode = if i.mode.kind == 1 { ProcessingMode::Streaming } else {
        ProcessingMode::FixedSamples { channels: i.mode.channels, sample_rate: i.mode.sample_rate, samples: i.mode.samples } };
    StreamInfo { mode, terminal: if i.terminal == 0 { None } else { Some(dec_err(i.terminal)) }, in_new: i.in_new != 0,
                 n_frames: i.n_frames, n_samples: i.n_samples }
}
