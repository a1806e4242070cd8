// hps.cpp -- HPR<GPU>, HPRRealtime<GPU>, HPRIOffline<GPU> over the C-ABI engine.
// Mirrors the control flow of the reference's libzen/hps.cu:21-427; the per-hop arithmetic
// (hps.cu:429-652) runs inside libzen_hip.so.
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <functional>
#include <iostream>
#include <numeric>
#include <system_error>
#include <exception>
#include <thread>
#include <utility>

#include <sys/mman.h>

#include <hps.h>
#include <libzen/hps.h>
#include <libzen/io.h>
#include <libzen/zen.h>

namespace zen {

void throw_or_die(int rc, const char* where)
{
	if (rc == ZEN_HIP_OK)
		return;
	if (rc == ZEN_HIP_E_FILTER_TOO_BIG)
		throw ZgException("median filter bigger than matrix dimension"); // mfilt.h:85
	if (rc == ZEN_HIP_E_HOPS_NOT_DIVISIBLE)
		throw ZgException("hop_h and hop_p should be evenly divisible"); // hps.cu:34
	if (rc == ZEN_HIP_E_BAD_ARG || rc == ZEN_HIP_E_UNSUPPORTED)
		throw ZgException(std::string(where) + ": " + zen_hip_last_error());
	std::cerr << where << ": " << zen_hip_last_error() << std::endl; // mfilt.h:166-215, io.h:37-66
	std::exit(1);
}

namespace internal {
	namespace hps {

		HPR<Backend::GPU>::HPR(float fs, std::size_t hop, float beta, unsigned int output_flags,
		                       mfilt::MedianFilterDirection causality, bool copy_bord, std::size_t max_hops)
		    : fs(fs)
		    , hop(hop)
		    , nwin(2 * hop)
		    , nfft(4 * hop)
		    , beta(beta)
		    , output_percussive((output_flags & zen::hps::OUTPUT_PERCUSSIVE) != 0)
		    , output_harmonic((output_flags & zen::hps::OUTPUT_HARMONIC) != 0)
		    , output_residual((output_flags & zen::hps::OUTPUT_RESIDUAL) != 0)
		    , use_sse(false)
		    , soft_mask(false)
		    , engine(nullptr)
		{
			throw_or_die(zen_hip_hpr_create(fs, hop, beta, output_flags, (int)causality, copy_bord ? 1 : 0, 1,
			                                max_hops, &engine),
			             "HPR");
			zen_hip_hpr_params p;
			throw_or_die(zen_hip_hpr_get_params(engine, &p), "HPR");
			l_harm = p.l_harm;
			l_perc = p.l_perc;
			lag = p.lag;
			stft_width = p.stft_width;
			COLA_factor = p.cola_factor;
		}

		HPR<Backend::GPU>::~HPR() { zen_hip_hpr_destroy(engine); }

		void HPR<Backend::GPU>::use_sse_filter()
		{
			use_sse = true;
			throw_or_die(zen_hip_hpr_use_sse_filter(engine), "use_sse_filter");
		}

		void HPR<Backend::GPU>::use_soft_mask()
		{
			soft_mask = true;
			throw_or_die(zen_hip_hpr_use_soft_mask(engine), "use_soft_mask");
		}

		void HPR<Backend::GPU>::reset_buffers() { throw_or_die(zen_hip_hpr_reset_buffers(engine), "reset_buffers"); }

		void HPR<Backend::GPU>::process_next_hop(InputPointer in_hop)
		{
			throw_or_die(zen_hip_hpr_process_next_hop(engine, in_hop.get()), "process_next_hop");
		}

		void HPR<Backend::GPU>::process_hops(InputPointer in, std::size_t n_hops, float* harm, float* perc,
		                                     float* resid)
		{
			throw_or_die(zen_hip_hpr_process(engine, in.get(), n_hops, n_hops * hop, harm, perc, resid, n_hops * hop),
			             "process_hops");
		}

		void HPR<Backend::GPU>::copy_out(unsigned int which, float* out_dev)
		{
			throw_or_die(zen_hip_hpr_copy_output(engine, which, out_dev), "copy_out");
		}

		std::vector<float> HPR<Backend::GPU>::snapshot(unsigned int which)
		{
			zen::internal::device_vector<float> tmp(hop);
			copy_out(which, tmp.raw());
			return tmp.to_host();
		}
		std::vector<float> HPR<Backend::GPU>::percussive_out() { return snapshot(zen::hps::OUTPUT_PERCUSSIVE); }
		std::vector<float> HPR<Backend::GPU>::harmonic_out() { return snapshot(zen::hps::OUTPUT_HARMONIC); }
		std::vector<float> HPR<Backend::GPU>::residual_out() { return snapshot(zen::hps::OUTPUT_RESIDUAL); }

	} // namespace hps
} // namespace internal

namespace hps {
	using zen::internal::hps::mfilt::MedianFilterDirection;
	typedef zen::internal::hps::HPR<Backend::GPU> HPRG;

	// ---- HPRRealtime<GPU> (reference libzen/hps.cu:282-427) --------------------------------------
	template <>
	HPRRealtime<Backend::GPU>::~HPRRealtime()
	{
		delete p_impl;
	}

	template <>
	HPRRealtime<Backend::GPU>::HPRRealtime(float fs, std::size_t hop, float beta, unsigned int output_flags)
	    : p_impl(new HPRG(fs, hop, beta, output_flags, MedianFilterDirection::TimeCausal, true))
	{
	}

	template <>
	HPRRealtime<Backend::GPU>::HPRRealtime(float fs, std::size_t hop, float beta, unsigned int output_flags,
	                                       bool nocopybord)
	    : p_impl(new HPRG(fs, hop, beta, output_flags, MedianFilterDirection::TimeCausal, !nocopybord))
	{
	}

	template <>
	HPRRealtime<Backend::GPU>::HPRRealtime(float fs, std::size_t hop, unsigned int output_flags)
	    : HPRRealtime(fs, hop, 2.0, output_flags)
	{
	}

	template <>
	HPRRealtime<Backend::GPU>::HPRRealtime(float fs, unsigned int output_flags)
	    : HPRRealtime(fs, 256, 2.0, output_flags)
	{
	}

	template <>
	void HPRRealtime<Backend::GPU>::use_sse_filter()
	{
		p_impl->use_sse_filter();
	}

	template <>
	void HPRRealtime<Backend::GPU>::use_soft_mask()
	{
		p_impl->use_soft_mask();
	}

	template <>
	void HPRRealtime<Backend::GPU>::process_next_hop(thrust::device_ptr<float> in_hop)
	{
		p_impl->process_next_hop(in_hop);
	}

	template <>
	void HPRRealtime<Backend::GPU>::copy_harmonic(thrust::device_ptr<float> out_hop)
	{
		p_impl->copy_out(OUTPUT_HARMONIC, out_hop.get());
	}

	template <>
	void HPRRealtime<Backend::GPU>::copy_percussive(thrust::device_ptr<float> out_hop)
	{
		p_impl->copy_out(OUTPUT_PERCUSSIVE, out_hop.get());
	}

	template <>
	void HPRRealtime<Backend::GPU>::copy_residual(thrust::device_ptr<float> out_hop)
	{
		p_impl->copy_out(OUTPUT_RESIDUAL, out_hop.get());
	}

	// The float* overloads are the CPU backend's in the reference (hps.cu:365-390).  On this backend a
	// raw pointer is taken to be device-accessible memory (device or mapped host), like device_ptr.
	template <>
	void HPRRealtime<Backend::GPU>::process_next_hop(float* in_hop)
	{
		p_impl->process_next_hop(thrust::device_pointer_cast(in_hop));
	}
	template <>
	void HPRRealtime<Backend::GPU>::copy_harmonic(float* out)
	{
		p_impl->copy_out(OUTPUT_HARMONIC, out);
	}
	template <>
	void HPRRealtime<Backend::GPU>::copy_percussive(float* out)
	{
		p_impl->copy_out(OUTPUT_PERCUSSIVE, out);
	}
	template <>
	void HPRRealtime<Backend::GPU>::copy_residual(float* out)
	{
		p_impl->copy_out(OUTPUT_RESIDUAL, out);
	}

	template <>
	void HPRRealtime<Backend::GPU>::process_hops(thrust::device_ptr<float> in, std::size_t n_hops,
	                                             thrust::device_ptr<float> harm, thrust::device_ptr<float> perc,
	                                             thrust::device_ptr<float> resid)
	{
		p_impl->process_hops(in, n_hops, harm.get(), perc.get(), resid.get());
	}

	template <>
	void HPRRealtime<Backend::GPU>::process_hops_host(const float* in, std::size_t n_hops, float* harm, float* perc, float* resid)
	{
		throw_or_die(zen_hip_hpr_process_host(p_impl->engine, in, n_hops, harm, perc, resid), "process_hops_host");
	}

	template <>
	void HPRRealtime<Backend::GPU>::use_resident_kernel(int idle_ms)
	{
		throw_or_die(zen_hip_hpr_set_resident(p_impl->engine, idle_ms), "use_resident_kernel");
	}

	template <>
	void HPRRealtime<Backend::GPU>::warmup(zen::io::IOGPU& io)
	{
		// reference hps.cu:392-408: 1000 hops of iota through the mapped buffers, then a reset
		const int test_iters = 1000;
		const std::size_t hop = p_impl->hop;
		std::vector<float> testdata(test_iters * hop);
		std::vector<float> outdata(test_iters * hop);
		std::iota(testdata.begin(), testdata.end(), 0.0F);
		for (int i = 0; i < test_iters; ++i) {
			std::copy(testdata.begin() + i * hop, testdata.begin() + (i + 1) * hop, io.host_in);
			p_impl->process_next_hop(io.device_in);
			p_impl->copy_out(OUTPUT_PERCUSSIVE, io.device_out.get()); // synchronises: host_in may be reused
			std::copy(io.host_out, io.host_out + hop, outdata.begin() + i * hop);
		}
		p_impl->reset_buffers();
		zen_hip_synchronize(nullptr);
	}

	template <>
	void HPRRealtime<Backend::GPU>::warmup()
	{
		zen::io::IOGPU io(p_impl->hop);
		warmup(io);
	}

	// ---- HPRIOffline<GPU> (reference libzen/hps.cu:21-221) ----------------------------------------
	static void* make_offline(float fs, std::size_t hop_h, std::size_t hop_p, float beta_h, float beta_p,
	                          bool nocopybord)
	{
		zen_hip_hpri_t e = nullptr;
		throw_or_die(zen_hip_hpri_create(fs, hop_h, hop_p, beta_h, beta_p, nocopybord ? 1 : 0, 1, &e), "HPRIOffline");
		return e;
	}

	template <>
	HPRIOffline<Backend::GPU>::~HPRIOffline()
	{
		zen_hip_hpri_destroy(static_cast<zen_hip_hpri_t>(engine));
	}

	template <>
	HPRIOffline<Backend::GPU>::HPRIOffline(float fs, std::size_t hop_h, std::size_t hop_p, float beta_h,
	                                       float beta_p, bool nocopybord)
	    : p_impl_h(nullptr)
	    , p_impl_p(nullptr)
	    , hop_h(hop_h)
	    , hop_p(hop_p)
	    , engine(make_offline(fs, hop_h, hop_p, beta_h, beta_p, nocopybord))
	{
	}

	template <>
	HPRIOffline<Backend::GPU>::HPRIOffline(float fs, std::size_t hop_h, std::size_t hop_p, float beta_h,
	                                       float beta_p)
	    : HPRIOffline(fs, hop_h, hop_p, beta_h, beta_p, false)
	{
	}

	template <>
	HPRIOffline<Backend::GPU>::HPRIOffline(float fs, std::size_t hop_h, std::size_t hop_p)
	    : HPRIOffline(fs, hop_h, hop_p, 2.0, 2.0)
	{
	}

	template <>
	HPRIOffline<Backend::GPU>::HPRIOffline(float fs)
	    : HPRIOffline(fs, 4096, 256, 2.0, 2.0)
	{
	}

	template <>
	void HPRIOffline<Backend::GPU>::use_sse_filter()
	{
		throw_or_die(zen_hip_hpri_use_sse_filter(static_cast<zen_hip_hpri_t>(engine)), "use_sse_filter");
	}

	template <>
	void HPRIOffline<Backend::GPU>::use_soft_mask()
	{
		throw_or_die(zen_hip_hpri_use_soft_mask(static_cast<zen_hip_hpri_t>(engine)), "use_soft_mask");
	}

	namespace {
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif
		// The signature hands back three fresh std::vector<float>(n): 12 bytes of never-touched memory per sample, which a
		// value-initialising constructor faults in one 4 KB page at a time on one thread (4.7 GB/s on the EPYC host of the
		// MI355X boxes: 410 ms per hour of audio, against 26 ms for the separation with its copies).  Same vectors, same
		// zeros, but the pages are asked for as transparent huge pages and populated by several threads before the
		// single-threaded zero fill runs over them: only advice to the kernel about memory the vector already owns; where
		// the advice is refused (old kernel, THP off) the constructor's path is all that happens.
		void populate(void* p, std::size_t bytes)
		{
			const std::uintptr_t page = 4096, huge = (std::uintptr_t)2 << 20;
			std::uintptr_t a = ((std::uintptr_t)p + page - 1) & ~(page - 1), e = ((std::uintptr_t)p + bytes) & ~(page - 1);
			if (e < a + 8 * huge)
				return;
			(void)madvise((void*)a, e - a, MADV_HUGEPAGE);
			unsigned k = std::thread::hardware_concurrency() / 8;
			k = k < 1 ? 1 : k > 8 ? 8 : k;
			const std::uintptr_t per = (((e - a) / k) + huge - 1) & ~(huge - 1);
			std::vector<std::thread> th;
			for (unsigned i = 0; i < k; ++i) {
				const std::uintptr_t b0 = a + i * per, b1 = b0 + per < e ? b0 + per : e;
				if (b0 >= b1)
					break;
				try {
					th.emplace_back([=] { (void)madvise((void*)b0, b1 - b0, MADV_POPULATE_WRITE); });
				}
				catch (...) { // no thread: the zero fill faults those pages in itself
					break;
				}
			}
			for (auto& t : th)
				t.join();
		}

		void fresh_zeros(std::vector<float>& v, std::size_t n)
		{
			v.reserve(n);
			populate(v.data(), n * sizeof(float));
			v.resize(n);
		}
	} // namespace

	namespace {
		// The pages of a vector's RESERVED capacity asked for as transparent huge pages and faulted in by helper threads that
		// nobody waits for until the end of the call: the ranges of the clip are appended behind them (below).  Only advice to
		// the kernel about memory the vector owns; where it is refused the appends fault the pages in themselves.
		struct Populator {
			std::vector<std::thread> th;
			void start(void* p, std::size_t bytes)
			{
				const std::uintptr_t page = 4096, huge = (std::uintptr_t)2 << 20;
				std::uintptr_t a = ((std::uintptr_t)p + page - 1) & ~(page - 1), e = ((std::uintptr_t)p + bytes) & ~(page - 1);
				if (e < a + 8 * huge)
					return;
				(void)madvise((void*)a, e - a, MADV_HUGEPAGE);
				unsigned k = std::thread::hardware_concurrency() / 8;
				k = k < 1 ? 1 : k > 6 ? 6 : k;
				// pieces dealt round robin, so that the helpers together advance through the vector from its start -- the order
				// in which the ranges arrive
				const std::uintptr_t piece = 16 * huge;
				for (unsigned i = 0; i < k; ++i) {
					try {
						th.emplace_back([=] {
							for (std::uintptr_t b0 = a + i * piece; b0 < e; b0 += k * piece)
								(void)madvise((void*)b0, (b0 + piece < e ? piece : e - b0), MADV_POPULATE_WRITE);
						});
					}
					catch (...) {
						break;
					}
				}
			}
			~Populator()
			{
				for (auto& t : th)
					if (t.joinable())
						t.join();
			}
		};

		// n zeros: the value-initialising resize runs BEHIND the helpers that fault the pages in (where it overtakes them it
		// faults the pages itself): one pass of the kernel's zeroing and one of the constructor's, overlapped, instead of one
		// after the other
		void fresh_zeros_overlapped(std::vector<float>& v, std::size_t n)
		{
			v.reserve(n);
			Populator p;
			p.start(v.data(), n * sizeof(float));
			v.resize(n);
		}

		struct AppendCtx {
			std::vector<float>* v[2];
			std::atomic<bool> out_of_order{false}; // (set from either output's thread)
		};
		// zen_hip_hpri_sink_fn: the range is appended to its vector (capacity reserved: no allocation, nothing to throw)
		extern "C" void append_range(void* user, int output, std::size_t begin, const float* samples, std::size_t count)
		{
			AppendCtx* c = static_cast<AppendCtx*>(user);
			std::vector<float>& v = *c->v[output];
			if (begin != v.size() || v.capacity() - v.size() < count) {
				c->out_of_order.store(true, std::memory_order_relaxed);
				return;
			}
			v.insert(v.end(), samples, samples + count);
		}
	} // namespace

	template <>
	std::array<std::vector<float>, 3> HPRIOffline<Backend::GPU>::process(std::vector<float> audio)
	{
		// return same-sized vectors as a result (hps.cu:131, :219-220)
		const std::size_t n = audio.size();
		static const bool trace = std::getenv("ZEN_TRACE_PROCESS") != nullptr; // stderr: where the wall time of a call goes
		static const bool plain = std::getenv("ZEN_PROCESS_PLAIN") != nullptr; // the path of rounds 4-5 (vectors first, then the call)
		const auto t0 = std::chrono::steady_clock::now();
		std::vector<float> harmonic_out, percussive_out, residual_out;
		if (n >= ((std::size_t)1 << 21) && !plain) {
			// Long clips (round 6).  The signature hands back three fresh std::vector<float>(n): 12 bytes of never-touched memory
			// per sample.  Built first and then filled (rounds 4-5) that was: fault the pages in, write zeros over them (the
			// value-initialising resize), pin them for the copies, and only then separate -- 50 + 20 + 26 ms per hour of audio,
			// one after the other.  Now the harmonic and percussive vectors only RESERVE their memory; the separation runs at
			// once and hands every finished range over from pinned staging memory of the engine
			// (zen_hip_hpri_process_sink), where it is appended to its vector -- written once, by a thread per output, while
			// the later ranges are still on the device; helper threads fault the reserved pages in ahead of the appends.  The
			// residual (all zeros: hps.cu:45-48, :200-204) is value-initialised on a thread of its own.
			harmonic_out.reserve(n);
			percussive_out.reserve(n);
			struct Worker {
				std::thread t;
				std::exception_ptr err;
				~Worker()
				{
					if (t.joinable())
						t.join();
				}
			} wr;
			bool resid_started = false;
			try {
				wr.t = std::thread([&wr, &residual_out, n] {
					try {
						fresh_zeros_overlapped(residual_out, n);
					}
					catch (...) {
						wr.err = std::current_exception();
					}
				});
				resid_started = true;
			}
			catch (const std::system_error&) {
			}
			int rc;
			AppendCtx ctx;
			ctx.v[0] = &harmonic_out;
			ctx.v[1] = &percussive_out;
			{
				Populator ph, pp;
				ph.start(harmonic_out.data(), n * sizeof(float));
				pp.start(percussive_out.data(), n * sizeof(float));
				rc = zen_hip_hpri_process_sink(static_cast<zen_hip_hpri_t>(engine), audio.data(), n, 1, 1, append_range, &ctx);
			}
			if (!resid_started)
				fresh_zeros_overlapped(residual_out, n);
			if (wr.t.joinable())
				wr.t.join();
			if (wr.err)
				std::rethrow_exception(wr.err);
			throw_or_die(rc, "HPRIOffline::process");
			if (ctx.out_of_order.load() || harmonic_out.size() != n || percussive_out.size() != n)
				throw ZgException("HPRIOffline::process: the ranges of the clip did not arrive in order (internal)");
		}
		else {
			if (n >= ((std::size_t)1 << 21)) { // ZEN_PROCESS_PLAIN (A/B: rounds 4-5): the three vectors first, populated and zeroed by threads
				std::thread t1, t2;
				std::exception_ptr e1, e2;
				struct Join {
					std::thread& t;
					~Join()
					{
						if (t.joinable())
							t.join();
					}
				} j1{t1}, j2{t2};
				bool s1 = false, s2 = false;
				try {
					t1 = std::thread([&] { try { fresh_zeros(harmonic_out, n); } catch (...) { e1 = std::current_exception(); } });
					s1 = true;
					t2 = std::thread([&] { try { fresh_zeros(percussive_out, n); } catch (...) { e2 = std::current_exception(); } });
					s2 = true;
				}
				catch (const std::system_error&) {
				}
				fresh_zeros(residual_out, n);
				if (t1.joinable())
					t1.join();
				if (t2.joinable())
					t2.join();
				if (!s1)
					fresh_zeros(harmonic_out, n);
				if (!s2)
					fresh_zeros(percussive_out, n);
				if (e1)
					std::rethrow_exception(e1);
				if (e2)
					std::rethrow_exception(e2);
			}
			harmonic_out.resize(n);
			percussive_out.resize(n);
			residual_out.resize(n);
			// residual_out stays as value-initialised: pass 2's residual is never written (hps.cu:45-48, :200-204),
			// so there is nothing to fetch for it
			if (n > 0)
				throw_or_die(zen_hip_hpri_process(static_cast<zen_hip_hpri_t>(engine), audio.data(), n,
				                                  harmonic_out.data(), percussive_out.data(), nullptr),
				             "HPRIOffline::process");
		}
		if (trace) {
			const auto t2 = std::chrono::steady_clock::now();
			zen_hip_hpri_host_stats st{};
			zen_hip_hpri_host_stats_get(static_cast<zen_hip_hpri_t>(engine), &st);
			std::cerr << "HPRIOffline::process: " << n << " samples, " << std::chrono::duration<double, std::milli>(t2 - t0).count()
			          << " ms (" << st.n_ranges << " ranges, setup " << st.setup_ms << " ms, enqueue " << st.enqueue_ms << " ms, pipeline "
			          << st.total_ms << " ms)" << std::endl;
		}
		// The by-value clip dies with this call: returning its pages to the system takes 25-60 ms per hour of audio (one
		// munmap of 635 MB), on the caller's clock.  A helper thread does it while the caller goes on.
		if (n >= ((std::size_t)1 << 23)) {
			try {
				std::thread([clip = std::move(audio)]() mutable { std::vector<float>().swap(clip); }).detach();
			}
			catch (const std::system_error&) { // no thread: the clip is freed on return, as before
			}
		}
		return std::array<std::vector<float>, 3>{std::move(harmonic_out), std::move(percussive_out),
		                                         std::move(residual_out)};
	}

	template class HPRIOffline<Backend::GPU>;
	template class HPRRealtime<Backend::GPU>;
} // namespace hps
} // namespace zen
