//! # QOA - Quite OK Audio Format, MI355X backend
//!
//! UNCOMPILED SOURCE (no Rust toolchain where this repository is built).  Same public items as
//! `rafaelcaricio/qoaudio` 1.0 (`src/lib.rs` of the reference; line numbers below refer to it), forwarded to
//! the C ABI in `include/qoaudio_amd.h`.  The reference forbids unsafe code; an FFI shim cannot.
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
    pub enum qoa_encoder {}
    pub enum qoa_decoder {}
    pub enum qoa_multi {}
    /// `int64_t (*)(void *ctx, uint8_t *buf, size_t len)`: bytes stored, 0 at the end of the stream, negative on error
    pub type qoa_read_fn = unsafe extern "C" fn(*mut c_void, *mut u8, usize) -> i64;
    extern "C" {
        pub fn qoa_last_error_detail() -> *const c_char;
        pub fn qoa_encoded_size(channels: u32, samples: u32) -> usize;
        pub fn qoa_encoder_new(desc: *const qoa_desc, out: *mut *mut qoa_encoder) -> c_int;
        pub fn qoa_encoder_free(enc: *mut qoa_encoder);
        pub fn qoa_encoder_encode(enc: *mut qoa_encoder, pcm: *const i16, n_i16: usize, out: *mut u8, cap: usize,
                                  written: *mut usize) -> c_int;
        pub fn qoa_encoder_write_header(enc: *const qoa_encoder, out: *mut u8) -> c_int;
        pub fn qoa_encoder_encode_frame(enc: *mut qoa_encoder, pcm: *const i16, n_i16: usize, out: *mut u8,
                                        cap: usize, written: *mut usize, frame_len: *mut usize) -> c_int;
        pub fn qoa_decoder_new(bytes: *const u8, len: usize, out: *mut *mut qoa_decoder) -> c_int;
        pub fn qoa_decoder_new_reader(read: qoa_read_fn, ctx: *mut c_void, out: *mut *mut qoa_decoder) -> c_int;
        pub fn qoa_decoder_new_streaming(out: *mut *mut qoa_decoder) -> c_int;
        pub fn qoa_decoder_free(dec: *mut qoa_decoder);
        pub fn qoa_decoder_mode(dec: *const qoa_decoder, mode: *mut qoa_mode) -> c_int;
        pub fn qoa_decoder_current_frame_header(dec: *const qoa_decoder, h: *mut qoa_frame_header) -> c_int;
        pub fn qoa_decoder_total_duration(dec: *const qoa_decoder) -> c_double;
        pub fn qoa_decoder_next(dec: *mut qoa_decoder, item: *mut qoa_item) -> c_int;
        pub fn qoa_decoder_decode_frame(dec: *mut qoa_decoder, frame: *const u8, len: usize, out: *mut i16,
                                        cap: usize, n_out: *mut usize) -> c_int;
        pub fn qoa_decode_all(bytes: *const u8, len: usize, channels: *mut u8, sample_rate: *mut u32,
                              out: *mut i16, cap: usize, n_out: *mut usize) -> c_int;
        // multi-GPU batches in one process (no counterpart in the reference; include/qoaudio_amd.h)
        pub fn qoa_multi_create(devices: *const c_int, n_devices: c_int, out: *mut *mut qoa_multi) -> c_int;
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
